#!/bin/bash
# MALL-residency probe for the streaming engine on the (16200,7200) code: per-codeword decode time at
# small batches (working set within the 256 MiB Infinity Cache) against the full batch, for the
# non-temporal-hint variants.  Output: gpurun_out/mall/probe.jsonl
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mall; mkdir -p $O
for lib in default nt_off ntl_off nts_off; do
  for b in 256 512 768 1024 2048 4096 32768; do
    if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
    timeout -k 10 120 python tools/time_sweeps.py --workload wrcq_dvbs2 --batch $b --tag $lib >> $O/probe.jsonl 2>> $O/probe.err || echo "fail $lib $b" >> $O/probe.err
  done
done
cat $O/probe.jsonl | cut -c1-300
