#!/bin/bash
# cn_gather tuning round: full GPU suite on the default build, then every load-group variant on config 5 and config 4
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gather2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for lib in default g2w5 g3w4 g4w3 g6w2 g8w2; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 200 python tools/time_sweeps.py --workload wrcq_dvbs2 --mode stream --tag $lib >> $O/time.jsonl 2>> $O/time.err
  timeout -k 10 200 python tools/time_sweeps.py --workload rcq --mode stream --tag $lib >> $O/time.jsonl 2>> $O/time.err
done
unset LDPC_HIP_LIB
cut -c1-330 $O/time.jsonl
timeout -k 10 300 python tools/f32_vs_f64.py > $O/f32_vs_f64.jsonl 2> $O/f32_vs_f64.err; cat $O/f32_vs_f64.jsonl
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_default.json
