#!/bin/bash
# parity suites + bench leg after the RCQ code-pair form became the streaming default
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pair3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 > $O/bench_wrcq.json 2> $O/bench_wrcq.err; cut -c1-1500 $O/bench_wrcq.json
timeout -k 10 300 python bench.py --workload rcq --steps 10 --warmup 3 > $O/bench_rcq.json 2> $O/bench_rcq.err; cut -c1-600 $O/bench_rcq.json
