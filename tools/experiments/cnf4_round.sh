#!/bin/bash
# cn_sweep_f4 tuning A/B on one box: default (5 waves/SIMD, non-temporal loads and stores) vs 6 waves, temporal loads, temporal stores
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03cnf4b; mkdir -p $O
for lib in default cnf4_w6 cnf4_ntl0 cnf4_nts0 default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 200 python tools/time_sweeps.py --workload basic --mode stream --tag $lib 2>> $O/time.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tag'], 'decode', round(d['decode_ms'], 3), 'cn', round(d['cn_ms'], 4), round(d['cn_GBs']), 'vn', round(d['vn_ms'], 4), round(d['vn_GBs']))" | tee -a $O/time.txt
done
unset LDPC_HIP_LIB
