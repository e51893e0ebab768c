cd "$GRAFT_REPO_ROOT"; O=gpurun_out/rep1; rm -rf $O; mkdir -p $O
for i in 1 2 3; do timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline > $O/w$i.json 2> $O/w$i.err; python -c "
import json,sys; d=json.loads(open('$O/w$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline'].get('measured_copy_GBps'))"; done
