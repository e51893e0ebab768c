#!/bin/bash
# callers of the path on the current build: Monte-Carlo driver throughput, one training step (8f-4), single-call latency
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03callers; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/time_simulator.py > $O/simulator.jsonl 2> $O/simulator.err; echo "simulator rc=$?"; cat $O/simulator.jsonl
timeout -k 10 300 python tools/time_train.py > $O/train.jsonl 2> $O/train.err; echo "train rc=$?"; cat $O/train.jsonl
timeout -k 10 300 python tools/time_single.py > $O/single.jsonl 2> $O/single.err; echo "single rc=$?"; cat $O/single.jsonl
