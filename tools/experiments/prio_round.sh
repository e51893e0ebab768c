#!/bin/bash
# resident kernel: wave priority by phase (s_setprio): 1 = variable phase (LDS-bound) high, 2 = check phase (VALU-bound) high
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prio; rm -rf $O; mkdir -p $O
for lib in default prio1 prio2 default prio1 prio2; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic rcq; do
    timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-stream-leg > $O/b.json 2> $O/b.err || echo "bench $lib $w failed"
    python3 -c "import json;d=json.load(open('$O/b.json'));print('$lib $w', round(d['ms_per_step'],4))" | tee -a $O/timings.txt
  done
done
