#!/bin/bash
# condense gpurun_out/r03 (tools/gpu_round.sh r03 bench|prof1|prof2|ctr) into profiles/
cd "$(dirname "$0")/.."
for w in basic neural2d rcq wrcq_dvbs2 basic_f64 rcq_layered; do python tools/summarize_rocprof.py r03 $w gpurun_out/r03/stats_$w gpurun_out/r03/fetch_$w gpurun_out/r03/write_$w | tail -1; done
python tools/summarize_rocprof.py r03 wrcq_dvbs2_gather gpurun_out/r03/stats_wrcq_dvbs2_gather | tail -1
rm -f profiles/counters.json
python tools/summarize_counters.py r03 basic resident_decode resident > /dev/null
python tools/summarize_counters.py r03 basic_f64 resident_decode resident_f64 > /dev/null
python tools/summarize_counters.py r03 wrcq_dvbs2 vn_sweep_q4 pair_vn > /dev/null
python tools/summarize_counters.py r03 wrcq_dvbs2 cn_sweep_q4 pair_cn > /dev/null
python tools/summarize_counters.py r03 rcq_layered layered_lds layered > /dev/null
mkdir -p profiles/bench
for f in default neural2d rcq wrcq_dvbs2 basic_f64 rcq_layered 2rank_gloo 2rank_gloo_wrcq_strong 1rank_rccl 2rank_torchrun; do [ -s gpurun_out/r03/bench_$f.json ] && cp gpurun_out/r03/bench_$f.json profiles/bench/r03_$f.json; done
ls profiles | grep r03 | wc -l
