set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TODA_GG_LDS_PF=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "subm or strided or determin or moments" > gpurun_out/t9.log 2>&1 || { tail -30 gpurun_out/t9.log; exit 1; }
tail -2 gpurun_out/t9.log
timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b9_a.json 2> gpurun_out/b9_a.err
TODA_GG_LDS_PF=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b9_b.json 2> gpurun_out/b9_b.err
timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b9_c.json 2> gpurun_out/b9_c.err
TODA_GG_LDS_PF=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b9_d.json 2> gpurun_out/b9_d.err
python - <<'PY'
import json
for f in "abcd":
    d=json.loads(open(f'gpurun_out/b9_{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
