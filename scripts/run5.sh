set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f -- python3 $R/bench.py --steps 5 --warmup 4 > /tmp/f_bench.log 2>&1
F=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_summary.py $F 5 $R/gpurun_out/f_timed.csv > $R/gpurun_out/f_groups.txt
python3 $R/toda_amd/tools/trace_gaps.py $F 5 > $R/gpurun_out/f_gaps.txt
python3 $R/toda_amd/tools/trace_sequence.py $F $R/gpurun_out/f_seq.txt
cat $R/gpurun_out/f_groups.txt; head -12 $R/gpurun_out/f_gaps.txt
