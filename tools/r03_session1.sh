#!/bin/bash
# round 3, GPU session 1: the whole -m gpu suite on the new build, then the layered timing
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s1; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest.log
timeout -k 10 120 python tools/time_layered.py > $O/layered.jsonl 2> $O/layered.err; echo "layered rc=$?" | tee -a $O/summary.txt
cat $O/layered.jsonl
