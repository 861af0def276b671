set -e
cd $GRAFT_REPO_ROOT
TODA_TORCH_PROFILE=gpurun_out/tp.txt timeout -k 10 400 python bench.py --steps 10 --warmup 5 > gpurun_out/tp_bench.json 2> gpurun_out/tp_bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python3 $R/bench.py --steps 3 --warmup 3 > /tmp/seq_bench.log 2>&1
F=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_sequence.py $F $R/gpurun_out/seq_c3.txt
python3 $R/toda_amd/tools/trace_gaps.py $F 3 > $R/gpurun_out/seq_gaps.txt
