#!/bin/bash
# threads per workgroup of the resident kernel: 768 threads (12 waves, 3 variable rounds instead of 4) need <= 80 VGPRs for two
# workgroups per CU (build lb768.so: launch bounds 768 / 6 waves per SIMD, probes build so LDPC_RESIDENT_NT applies)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/nt1; mkdir -p $O
for w in basic rcq neural2d; do
  timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag default >> $O/time.jsonl 2>> $O/time.err
  for nt in 512 640 768; do
    LDPC_HIP_LIB=$PWD/build_variants/lb768.so LDPC_RESIDENT_NT=$nt timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag lb768_nt$nt >> $O/time.jsonl 2>> $O/time.err
  done
done
cut -c1-330 $O/time.jsonl
timeout -k 10 100 python tools/time_single.py > $O/single.jsonl 2>> $O/time.err; cat $O/single.jsonl
