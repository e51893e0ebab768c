#!/bin/bash
# vn_sweep_q4: rows of two variables in flight per wait (LDPC_VNQ_PAIR) at 8 / 6 (default) / 5 waves per SIMD against one variable per wait
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/vnqpair; rm -rf $O; mkdir -p $O
for lib in nopair default pair_w8 pair_w5 nopair default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline --no-legs > $O/bench_$lib.json 2> $O/bench_$lib.err || echo "bench $lib failed"
  python3 - "$O/bench_$lib.json" $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(f'{sys.argv[2]:10s} step {d["ms_per_step"]:.3f} ms  vn {r["ms_per_launch"]:.4f}  cn {r["cn_sweep_q4"]["ms_per_launch"]:.4f}')
PY
done
unset LDPC_HIP_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "key_float or code_pair or wrcq or rcq or random_graphs" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
