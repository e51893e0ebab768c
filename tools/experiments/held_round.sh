#!/bin/bash
# resident kernel on top of res_select4: register-held check form (LDPC_RES_CHECK_MODE=1), plan prefetch with 32-bit offsets (LDPC_RES_PLAN_U32=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/held; rm -rf $O; mkdir -p $O
for lib in default held planu32 heldu32 default held planu32 heldu32; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic neural2d rcq basic_f64; do
    timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-stream-leg > $O/b.json 2> $O/b.err || echo "bench $lib $w failed"
    python3 -c "import json;d=json.load(open('$O/b.json'));print('$lib $w', round(d['ms_per_step'],4))" | tee -a $O/timings.txt
  done
done
