set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_conv2d.py -x -q > gpurun_out/t8.log 2>&1 || { tail -30 gpurun_out/t8.log; exit 1; }
tail -2 gpurun_out/t8.log
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python -m toda_amd.tools.bench_conv2d --config c3 > gpurun_out/conv2d_c3.jsonl 2> gpurun_out/conv2d_c3.err
python - <<'PY'
import json
for l in open('gpurun_out/conv2d_c3.jsonl'):
    r=json.loads(l)
    if 'layer' in r: print(r['layer'], r['fwd_ms'], r['dgrad_ms'], r['wgrad_ms'], '| miopen', r['miopen_fwd_ms'], r['miopen_dgrad_ms'], r['miopen_wgrad_ms'])
    else: print(r)
PY
