#!/bin/bash
# Recipe of the round-5 evidence in profiles/ (run on the MI355X box from the repository root, one part per gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r05.sh traces'      (kernel traces + groups, C3 / C5, both matrix paths)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r05.sh counters'    (PMC passes for roofline.traffic, both paths)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r05.sh lines'       (bench lines of record)
# Everything is written to gpurun_out/r05_*; the files worth judging are copied into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
PART=${1:-traces}
if [ "$PART" = "traces" ]; then
cd /tmp && export TMPDIR=/tmp
for mm in split native; do
  for w in c3 c5; do
    rm -rf /tmp/r05_kt_${w}_$mm
    TODA_MM=$mm timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_kt_${w}_$mm -- python3 $R/bench.py --workload $w --steps 5 --warmup 4 --no-cpu-baseline > $O/r05_kt_${w}_$mm.log 2>&1
    F=$(find /tmp/r05_kt_${w}_$mm -name "*kernel_trace.csv" | head -1)
    cp $(find /tmp/r05_kt_${w}_$mm -name "*kernel_stats.csv" | head -1) $O/r05_bench_${w}_${mm}_kernel_stats.csv
    python3 $R/toda_amd/tools/trace_summary.py $F 5 $O/r05_bench_${w}_${mm}_timed_steps.csv > $O/r05_bench_${w}_${mm}_groups.txt
    python3 $R/toda_amd/tools/trace_by_shape.py $F 5 $O/r05_sparse_conv_by_launch_shape_${w}_$mm.csv > /dev/null || true
    echo "  $w $mm done"
  done
done
fi
if [ "$PART" = "counters" ]; then
cd /tmp && export TMPDIR=/tmp
pmc() {   # pmc <tag> <counters> <bench args...>: one rocprofv3 pass over bench.py
  local tag=$1 c=$2; shift 2
  rm -rf /tmp/r05_pmc_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r05_pmc_$tag -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/r05_pmc_$tag.log 2>&1 || echo "pass $tag failed"
}
for mm in split native; do
  export TODA_MM=$mm
  for w in c3 c5; do
    pmc ${w}_${mm}_FETCH FETCH_SIZE --workload $w --steps 2 --warmup 2
    pmc ${w}_${mm}_WRITE WRITE_SIZE --workload $w --steps 2 --warmup 2
    pmc ${w}_${mm}_BUSY "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" --workload $w --steps 2 --warmup 2
    echo "  $w $mm done"
  done
done
export TODA_MM=split
# the other lines of record: the SAME step counts as the lines (their batches, hence their launch shapes, follow the step index)
pmc c2_split_FETCH FETCH_SIZE --workload c2 --steps 2 --warmup 2
pmc c2_split_WRITE WRITE_SIZE --workload c2 --steps 2 --warmup 2
for w in c5mix c5cl; do
  pmc ${w}_split_FETCH FETCH_SIZE --workload $w --steps 50 --warmup 40
  pmc ${w}_split_WRITE WRITE_SIZE --workload $w --steps 50 --warmup 40
  echo "  $w done"
done
S3=$(find /tmp/r05_pmc_c3_split_* /tmp/r05_pmc_c2_split_* -name "*counter_collection.csv"); N3=$(find /tmp/r05_pmc_c3_native_* -name "*counter_collection.csv")
S5=$(find /tmp/r05_pmc_c5_split_* /tmp/r05_pmc_c5mix_split_* /tmp/r05_pmc_c5cl_split_* -name "*counter_collection.csv"); N5=$(find /tmp/r05_pmc_c5_native_* -name "*counter_collection.csv")
python3 $R/toda_amd/tools/pmc_summary.py "gg_split_kernel<2, 1, 4, 2" $O/r05_pmc_split_64x64.json $S3 > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "gg_split_kernel<1, 1, 2, 2" $O/r05_pmc_split_32x32.json $S3 > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "gg_split_kernel<4, 1, 8, 2" $O/r05_pmc_split_128x128.json $S5 > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "wgrad_split_kernel" $O/r05_pmc_split_wgrad.json $S3 $(find /tmp/r05_pmc_c5_split_* -name "*counter_collection.csv") > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "gather_gemm_lds_kernel<4, 4, 2" $O/r05_pmc_gather_gemm_64x64.json $N3 > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "gather_gemm_lds_kernel<2, 2, 2" $O/r05_pmc_gather_gemm_32x32.json $N3 > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py "gather_gemm_lds_kernel<8, 8, 1" $O/r05_pmc_gather_gemm_128x128.json $N5 > /dev/null
# Winograd kernels, ONE layer shape per process (persistent kernels: the grid is the CU count whatever the shape)
for sh in 0 1; do
  tag=$([ $sh = 0 ] && echo 256x94 || echo 128x188)
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    t=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/r05_pw_${tag}_$t
    WINO_SHAPE=$sh N_IT=10 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r05_pw_${tag}_$t -- python3 $R/profiles/scripts/wino_bench.py > $O/r05_pw_${tag}_$t.log 2>&1 || echo "wino pass failed"
  done
  W=$(find /tmp/r05_pw_${tag}_* -name "*counter_collection.csv")
  python3 $R/toda_amd/tools/pmc_summary.py wino_fwd_ws_kernel $O/r05_pmc_wino_fwd_$tag.json $W > /dev/null
  python3 $R/toda_amd/tools/pmc_summary.py wino_wgrad_kernel $O/r05_pmc_wino_wgrad_$tag.json $W > /dev/null
done
fi
if [ "$PART" = "lines" ]; then
cd $R
export TODA_BENCH_STEP_MS=1
timeout -k 10 400 python bench.py --steps 50 --warmup 40 --layers --layers-out $O/r05_layers_c3.json > $O/r05_bench_c3.json 2> $O/r05_bench_c3.err
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_c3_driver_shape.json 2> $O/r05_bench_c3_driver_shape.err
TODA_MM=native timeout -k 10 400 python bench.py --steps 50 --warmup 40 --no-cpu-baseline --layers --layers-out $O/r05_layers_c3_native.json > $O/r05_bench_c3_native.json 2> $O/r05_bench_c3_native.err
for w in c2 c5 c5mix c5cl; do
  timeout -k 10 400 python bench.py --workload $w --steps 50 --warmup 40 > $O/r05_bench_$w.json 2> $O/r05_bench_$w.err
  echo "  $w done"
done
TODA_MM=native timeout -k 10 400 python bench.py --workload c5 --steps 50 --warmup 40 --no-cpu-baseline > $O/r05_bench_c5_native.json 2> $O/r05_bench_c5_native.err
timeout -k 10 400 python bench.py --workload c5 --steps 20 --warmup 8 --no-cpu-baseline --layers --layers-out $O/r05_layers_c5.json > /dev/null 2> $O/r05_layers_c5.err
timeout -k 10 400 python bench.py --steps 300 --warmup 40 --no-cpu-baseline > $O/r05_soak_c3.json 2> $O/r05_soak_c3.err
fi
echo "== done $PART"
