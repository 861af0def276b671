set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q > gpurun_out/t13.log 2>&1 || { tail -40 gpurun_out/t13.log; exit 1; }
tail -2 gpurun_out/t13.log
for v in 1 0 1 0; do
TODA_CLASS_DGRAD=$v timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b13_$v.json 2> gpurun_out/b13_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b13_$v.json').read().strip().splitlines()[-1])
print($v, d['value'], d['ms_per_step_median'], d['host_cpu_ms_per_step'])
PY
done
for v in 1 0; do
TODA_CLASS_DGRAD=$v timeout -k 10 300 python bench.py --workload c5 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b13c5_$v.json 2> gpurun_out/b13c5_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b13c5_$v.json').read().strip().splitlines()[-1])
print('c5', $v, d['value'], d['ms_per_step_median'])
PY
done
