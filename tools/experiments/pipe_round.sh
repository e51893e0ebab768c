#!/bin/bash
# resident kernel, check phase pass 1: next group of four edges requested before the current one is absorbed (LDPC_RES_PIPE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pipe; rm -rf $O; mkdir -p $O
for lib in nopipe default nopipe default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic neural2d rcq; do
    timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-stream-leg > $O/b.json 2> $O/b.err || echo "bench $lib $w failed"
    python3 -c "import json;d=json.load(open('$O/b.json'));print('$lib $w', round(d['ms_per_step'],4))" | tee -a $O/timings.txt
  done
done
unset LDPC_HIP_LIB
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not bench and (auto or resident)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
