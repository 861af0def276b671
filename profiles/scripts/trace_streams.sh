#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; W=${1:-c3}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tq
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tq -- python3 $R/bench.py --workload $W --steps 5 --warmup 4 --no-cpu-baseline > $O/ts_$W.log 2>&1
F=$(find /tmp/tq -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/scripts/trace_streams.py $F 5 $R > $O/ts_${W}_streams.txt
TS_QUEUE_RANK=1 python3 $R/profiles/scripts/trace_streams.py $F 5 $R > $O/ts_${W}_side.txt
python3 $R/toda_amd/tools/trace_summary.py $F 5 $O/ts_${W}_timed_steps.csv > $O/ts_${W}_groups.txt
wc -l $O/ts_${W}_streams.txt
