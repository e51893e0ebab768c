#!/bin/bash
# cn_gather: XCD-affine tile mapping and 64-codeword tiles (L2 residency of a tile's rows), probes build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/xcd1; mkdir -p $O
export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/probes.so
for cfg in "0 4" "1 4" "0 1" "1 1"; do
  set -- $cfg
  for w in wrcq_dvbs2 rcq; do
    LDPC_GATHER_XCD=$1 LDPC_STREAM_VEC=$2 timeout -k 10 300 python tools/time_sweeps.py --workload $w --mode stream --tag "xcd$1_vec$2" >> $O/time.jsonl 2>> $O/time.err
  done
done
cut -c1-330 $O/time.jsonl
