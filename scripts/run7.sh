set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python3 $R/bench.py --workload c5 --steps 6 --warmup 4 > /tmp/g_bench.log 2>&1
F=$(find /tmp/prof_g -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_summary.py $F 6 $R/gpurun_out/g_timed.csv > $R/gpurun_out/g_groups.txt
python3 $R/toda_amd/tools/trace_gaps.py $F 6 > $R/gpurun_out/g_gaps.txt
python3 $R/toda_amd/tools/trace_by_shape.py $F > $R/gpurun_out/g_shapes.txt 2>&1 || true
head -24 $R/gpurun_out/g_groups.txt; head -3 $R/gpurun_out/g_gaps.txt
