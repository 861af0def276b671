set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1 || { tail -40 gpurun_out/t12.log; exit 1; }
tail -2 gpurun_out/t12.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b12_$i.json 2> gpurun_out/b12_$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/b12_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_median'], d['host_cpu_ms_per_step'], d['host_issue_ms_per_step'])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt12 -- python3 $R/bench.py --steps 5 --warmup 4 --no-cpu-baseline > /tmp/kt12.log 2>&1
F=$(find /tmp/kt12 -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_summary.py $F 5 $R/gpurun_out/kt12_timed.csv | head -3
python3 $R/toda_amd/tools/trace_gaps.py $F 5 | head -2
