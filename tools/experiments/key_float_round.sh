#!/bin/bash
# float form of the code-pair key (key_pair4) on the shipped build: exactness test, the code-pair parity tests, config 5 timing
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/keyfloat; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "key_float or code_pair or wrcq or rcq or random_graphs" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_wrcq.json 2> $O/bench_wrcq.err && cut -c1-900 $O/bench_wrcq.json
