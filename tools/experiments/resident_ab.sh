#!/bin/bash
# same-box A/B: current default build vs the round-start build (old.so), wide-check timing, full GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/res5; mkdir -p $O
for lib in default old default old; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic rcq; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
cut -c1-140 $O/time.jsonl
timeout -k 10 200 python tools/time_wide.py >> $O/wide.jsonl 2>> $O/wide.err
LDPC_HIP_LIB=$PWD/build_variants/nowide.so timeout -k 10 200 python tools/time_wide.py >> $O/wide.jsonl 2>> $O/wide.err
cat $O/wide.jsonl; tail -3 $O/wide.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 100 python tools/time_single.py > $O/single.jsonl 2>> $O/time.err; cat $O/single.jsonl
