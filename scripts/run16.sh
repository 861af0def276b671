set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/abl2.txt
for a in 0 512 1536 16 0; do TODA_WINO_ABLATE=$a PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 120 python scripts/abl.py >> gpurun_out/abl2.txt 2>> gpurun_out/abl2.err; done
cat gpurun_out/abl2.txt
