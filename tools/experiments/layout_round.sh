#!/bin/bash
# timing-only (results are WRONG in the variant builds): V2C codes of the code-pair form in variable-major order -- the variable sweep
# then writes contiguous rows (LDPC_EXP_LAYOUT bit 0), the check sweep gathers its rows from scattered positions (bit 1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/layout; rm -rf $O; mkdir -p $O
for lib in default lay_vn lay_cn lay_both default lay_both; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs > $O/bench_$lib.json 2> $O/bench_$lib.err || echo "bench $lib failed"
  python3 - "$O/bench_$lib.json" $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(f'{sys.argv[2]:10s} step {d["ms_per_step"]:.3f} ms  vn {r["ms_per_launch"]:.4f}  cn {r["cn_sweep_q4"]["ms_per_launch"]:.4f}')
PY
done
