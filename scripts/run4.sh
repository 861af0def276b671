set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_center_loss.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -x -q > gpurun_out/t4.log 2>&1 || { tail -40 gpurun_out/t4.log; exit 1; }
tail -3 gpurun_out/t4.log
timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b4_a.json 2> gpurun_out/b4_a.err
TODA_FUSED_LOSS=0 timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b4_b.json 2> gpurun_out/b4_b.err
cut -c1-200 gpurun_out/b4_a.json; cut -c1-200 gpurun_out/b4_b.json
