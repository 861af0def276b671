set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1 || { tail -40 gpurun_out/t6.log; exit 1; }
tail -3 gpurun_out/t6.log
for w in c3 c5 c5mix c5cl c2; do timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 8 > gpurun_out/b6_$w.json 2> gpurun_out/b6_$w.err; cut -c1-160 gpurun_out/b6_$w.json; done
