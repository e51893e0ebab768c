#!/bin/bash
# One GPU-box session for the tracked evidence of a round: bench lines for every workload, rocprofv3 kernel stats and
# the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, counters only with --kernel-trace) for all four workloads
# WITH the streaming-engine leg on, SQ counters of the two dominant kernels.  Outputs under gpurun_out/<tag>/;
# tools/summarize_rocprof.py condenses them into profiles/.
#   bash tools/gpu_round.sh r02
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "default bench failed" >> $O/errors.log
for w in neural2d rcq wrcq_dvbs2 basic_f64; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed" >> $O/errors.log
done
# 2-rank rehearsal of the N > 1 path on this one GPU, started BARE (bench.py spawns its ranks); gloo stands in for RCCL
LDPC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline --no-stream-leg > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || echo "2-rank rehearsal failed" >> $O/errors.log
LDPC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload wrcq_dvbs2 --strong --batch 16384 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_wrcq_strong.json 2> $O/bench_2rank_gloo_wrcq.err || echo "2-rank wrcq rehearsal failed" >> $O/errors.log
timeout -k 10 300 python bench.py --gpus 1 --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err || echo "1-rank RCCL failed" >> $O/errors.log
echo "bench lines done"
for w in basic neural2d rcq wrcq_dvbs2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats_$w.log 2>&1 || echo "stats $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/fetch_$w.log 2>&1 || echo "fetch $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/write_$w.log 2>&1 || echo "write $w failed" >> $O/errors.log
  echo "profiled $w"
done
# the fused gather form of config 5 (the code-pair form is the default there): same three passes with the form forced
export LDPC_ENGINE_MODE=gather
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wrcq_dvbs2_gather -- python3 bench.py --workload wrcq_dvbs2 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats_wrcq_dvbs2_gather.log 2>&1 || echo "stats gather failed" >> $O/errors.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_wrcq_dvbs2_gather -- python3 bench.py --workload wrcq_dvbs2 --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/fetch_wrcq_dvbs2_gather.log 2>&1 || echo "fetch gather failed" >> $O/errors.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_wrcq_dvbs2_gather -- python3 bench.py --workload wrcq_dvbs2 --steps 2 --warmup 1 --no-cpu-baseline --no-legs --sweep-reps 3 > $O/write_wrcq_dvbs2_gather.log 2>&1 || echo "write gather failed" >> $O/errors.log
unset LDPC_ENGINE_MODE
# SQ / LDS counters of the two dominant kernels (three passes each)
N=1
pass() { local w=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/ctr_${w}_p$N -- python3 tools/time_sweeps.py --workload $w > $O/ctr_${w}_p$N.log 2>&1 || echo "counter pass $w $N failed" >> $O/errors.log; N=$((N+1)); }
for w in basic wrcq_dvbs2; do
  N=1
  pass $w SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES
  pass $w SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
  pass $w GRBM_COUNT GRBM_GUI_ACTIVE
done
# per-phase instruction histogram of the resident kernel: the probes build skips phases (LDPC_RES_DEBUG bits: 1 check, 2 variable,
# 4 output, 8 final posterior+syndrome, 16 LLR load, 32 init); counter differences against skip=0 give each phase's share
export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/probes.so
for d in 0 1 2 4 8 16 32; do
  export LDPC_RES_DEBUG=$d
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/phase_skip$d -- python3 tools/time_sweeps.py --workload basic > $O/phase_skip$d.log 2>&1 || echo "phase pass $d failed" >> $O/errors.log
done
unset LDPC_HIP_LIB LDPC_RES_DEBUG
cat $O/bench_*.json | cut -c1-300; cat $O/errors.log 2>/dev/null; ls $O | head -50
