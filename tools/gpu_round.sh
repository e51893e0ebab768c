#!/bin/bash
# GPU-box sessions for the tracked evidence of a round (one gpurun call per PART; a call is limited to 20 minutes):
#   bench : bench.py lines -- default (headline + legs), every workload alone, the multi-rank lines (2 ranks over gloo on this one
#           GPU, 1 rank over RCCL), all started BARE
#   prof1 / prof2 : rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, counters only with
#           --kernel-trace) per workload, stream leg ON;  prof2 also the fused-gather form of config 5
#   ctr   : SQ / LDS counters of the dominant kernels (three passes each)
# Outputs under gpurun_out/<tag>/; tools/summarize_rocprof.py / summarize_counters.py condense them into profiles/.
#   bash tools/gpu_round.sh r03 bench
set -o pipefail
TAG=${1:-r03}
PART=${2:-bench}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
prof() {   # stats + fetch + write for one workload
  local w=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats_$w.log 2>&1 || echo "stats $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/fetch_$w.log 2>&1 || echo "fetch $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/write_$w.log 2>&1 || echo "write $w failed" >> $O/errors.log
  echo "profiled $w"
}
case $PART in
bench)
  timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "default bench failed" >> $O/errors.log
  for w in neural2d rcq wrcq_dvbs2 basic_f64 rcq_layered; do
    timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed" >> $O/errors.log
  done
  # the DEFAULT multi-rank line (what the driver's `bench.py --gpus N` prints): config 2 weak + config 5 sharded weak and strong (the
  # strong leg cut to 16384 codewords in total: two ranks share this one GPU), rank inventory, all-gather alone, content check, CPU baseline
  LDPC_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --leg-steps 2 --config5-total 16384 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || echo "2-rank rehearsal failed" >> $O/errors.log
  LDPC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload wrcq_dvbs2 --strong --batch 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_wrcq_strong.json 2> $O/bench_2rank_gloo_wrcq.err || echo "2-rank wrcq rehearsal failed" >> $O/errors.log
  timeout -k 10 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 2 --leg-steps 2 --no-cpu-baseline --no-stream-leg > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err || echo "1-rank RCCL failed" >> $O/errors.log
  # the driver's own launch form for N > 1 (torch.distributed.run; gloo stands in for RCCL on this one GPU)
  LDPC_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --leg-steps 1 --config5-total 8192 --no-cpu-baseline --no-stream-leg 2> $O/bench_2rank_torchrun.err | grep '^{' > $O/bench_2rank_torchrun.json || echo "2-rank torchrun rehearsal failed" >> $O/errors.log
  timeout -k 10 120 python tools/time_layered.py > $O/layered.jsonl 2> $O/layered.err || echo "layered timing failed" >> $O/errors.log
  echo "bench lines done"; cat $O/bench_*.json | cut -c1-260
  ;;
prof1) for w in basic neural2d rcq; do prof $w; done ;;
prof2)
  for w in wrcq_dvbs2 basic_f64 rcq_layered; do prof $w; done
  export LDPC_ENGINE_MODE=gather        # the fused gather form of config 5 (the code-pair form is the default there)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wrcq_dvbs2_gather -- python3 bench.py --workload wrcq_dvbs2 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats_wrcq_dvbs2_gather.log 2>&1 || echo "stats gather failed" >> $O/errors.log
  unset LDPC_ENGINE_MODE
  ;;
ctr)
  N=1
  pass() { local w=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/ctr_${w}_p$N -- python3 tools/time_sweeps.py --workload $w > $O/ctr_${w}_p$N.log 2>&1 || echo "counter pass $w $N failed" >> $O/errors.log; N=$((N+1)); }
  for w in basic wrcq_dvbs2 basic_f64 rcq_layered; do
    N=1
    pass $w SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES
    pass $w SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
    pass $w GRBM_COUNT GRBM_GUI_ACTIVE
    echo "counters $w"
  done
  ;;
esac
cat $O/errors.log 2>/dev/null; ls $O | wc -l
