set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_conv2d.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -x -q > gpurun_out/t18.log 2>&1 || { tail -40 gpurun_out/t18.log; exit 1; }
tail -2 gpurun_out/t18.log
for v in 1 0 1 0; do
TODA_DECONV_AS_CONV1X1=$v timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b18_$v.json 2> gpurun_out/b18_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b18_$v.json').read().strip().splitlines()[-1])
print($v, d['value'], d['ms_per_step_median'])
PY
done
