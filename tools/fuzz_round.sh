#!/bin/bash
# randomized differential campaign: the random-graph property test (resident == stream == sweeps == CPU oracle, every decoder
# family, random degrees incl. wide checks, random T / batch / weights / stop mode) over hundreds of seeds
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03fuzz; mkdir -p $O
LDPC_FUZZ_SEEDS=${1:-400} timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -q -x -k "random_graphs and auto" > $O/fuzz.log 2>&1; echo "rc=$?" >> $O/fuzz.log
tail -5 $O/fuzz.log
