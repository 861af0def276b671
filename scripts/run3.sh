set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TODA_WINO_ABLATE=256 timeout -k 10 300 python -m pytest tests/test_gpu_conv2d.py -x -q > gpurun_out/t3.log 2>&1 || { tail -30 gpurun_out/t3.log; exit 1; }
tail -2 gpurun_out/t3.log
for a in 0 256 1 2 257 258 4 260 7 263; do TODA_WINO_ABLATE=$a PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 120 python scripts/abl.py >> gpurun_out/abl.txt 2>> gpurun_out/abl.err; done
cat gpurun_out/abl.txt
