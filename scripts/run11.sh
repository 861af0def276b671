set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in x16 x32; do
export TODA_HIP_LIB=$R/scratch_build/libtoda_$v.so
rm -rf /tmp/pmc_f
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
echo $v
python3 $R/toda_amd/tools/pmc_summary.py gather_gemm_lds_kernel $R/gpurun_out/pmc_f_$v.json $(find /tmp/pmc_f -name "*counter_collection.csv") | grep "4, 4, 2"
done
cd $R
for v in x16 x32 x16 x32; do
  export TODA_HIP_LIB=$R/scratch_build/libtoda_$v.so
  timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/b11_$v.json 2> gpurun_out/b11_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b11_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
done
