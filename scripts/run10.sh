set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/t10.log 2>&1 || { tail -30 gpurun_out/t10.log; exit 1; }
tail -2 gpurun_out/t10.log
for v in def x0 x16 x256 def x0; do
  if [ $v = def ]; then unset TODA_HIP_LIB; else export TODA_HIP_LIB=$R/scratch_build/libtoda_$v.so; fi
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b10_$v.json 2> gpurun_out/b10_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b10_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
done
unset TODA_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
python3 $R/toda_amd/tools/pmc_summary.py gather_gemm_lds_kernel $R/gpurun_out/pmc_f.json $(find /tmp/pmc_f -name "*counter_collection.csv") | grep "4, 4, 2"
