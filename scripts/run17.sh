set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c3 c5 c5cl c2; do
timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b17_$w.json 2> gpurun_out/b17_$w.err || { tail -5 gpurun_out/b17_$w.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/b17_$w.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$w', d['value'], d['ms_per_step_median'], r['frac'], r['avg_launch_ms'], r['kernel'], r.get('traffic'))
PY
done
