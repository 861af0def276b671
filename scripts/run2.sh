set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_conv2d.py tests/test_gpu_e2e.py tests/test_gpu_golden.py -x -q > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b2_a.json 2> gpurun_out/b2_a.err
TODA_PLANES_BN=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b2_b.json 2> gpurun_out/b2_b.err
timeout -k 10 300 python bench.py --steps 30 --warmup 8 > gpurun_out/b2_c.json 2> gpurun_out/b2_c.err
cut -c1-330 gpurun_out/b2_a.json; cut -c1-330 gpurun_out/b2_b.json; cut -c1-330 gpurun_out/b2_c.json
