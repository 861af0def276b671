set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t15.log 2>&1 || { tail -40 gpurun_out/t15.log; exit 1; }
tail -2 gpurun_out/t15.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b15_$i.json 2> gpurun_out/b15_$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b15_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_median'], d['host_cpu_ms_per_step'], d['host_issue_ms_per_step'])
PY
done
TODA_PREFETCH=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b15_p.json 2> gpurun_out/b15_p.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b15_p.json').read().strip().splitlines()[-1])
print('prefetch', d['value'], d['ms_per_step_median'], d['host_cpu_ms_per_step'], d['host_issue_ms_per_step'])
PY
