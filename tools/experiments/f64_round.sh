#!/bin/bash
# float64 resident kernel: branch-free min1/min2 (v_min/v_max_f64), products hoisted out of the edge loop, pass 2 in groups of four edges
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f64mm; rm -rf $O; mkdir -p $O
for lib in nof64mm default nof64mm default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic_f64 basic; do
    timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-stream-leg > $O/b.json 2> $O/b.err || echo "bench $lib $w failed"
    python3 -c "import json;d=json.load(open('$O/b.json'));print('$lib $w', round(d['ms_per_step'],4))" | tee -a $O/timings.txt
  done
done
unset LDPC_HIP_LIB
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_simulation_framework.py -x -q -m gpu -k "not bench and (auto or resident or fp64 or f64 or golden or simulator)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
