#!/bin/bash
# Recipe of the round-4 evidence in profiles/ (run on the MI355X box from the repository root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r04.sh lines'       (bench lines, soak)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r04.sh counters'    (PMC passes, per-layer dense convs)
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r04.sh traces'      (kernel traces, gaps, stall counters)
# (three calls: one gpurun call is limited to 20 minutes)
# Everything is written to gpurun_out/r04_*; the files worth judging are copied into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
PART=${1:-all}
if [ "$PART" = "lines" ] || [ "$PART" = "all" ]; then
echo "== bench lines"
# 40 warm-up steps: on these boxes a power-management transient 0.6-0.7 s after the first step (steps 33-35 of a run) costs 3-12 ms on
# 3-8 consecutive steps (per-step times in the .err files, TODA_BENCH_STEP_MS); the *_driver_shape line is the driver's K / W
export TODA_BENCH_STEP_MS=1
timeout -k 10 400 python bench.py --steps 50 --warmup 40 --layers --layers-out $O/r04_layers_c3.json > $O/r04_bench_c3.json 2> $O/r04_bench_c3.err
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_c3_driver_shape.json 2> $O/r04_bench_c3_driver_shape.err
# the same line with the whole process held on two host cores (the share of a rank on an 8-GPU node of 16 cores): plain launch, no profiler
timeout -k 10 400 taskset -c 0,1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_c3_driver_shape_2cores.json 2> $O/r04_bench_c3_driver_shape_2cores.err
for w in c2 c5 c5mix c5cl; do
  timeout -k 10 400 python bench.py --workload $w --steps 50 --warmup 40 > $O/r04_bench_$w.json 2> $O/r04_bench_$w.err
  echo "  $w done"
done
timeout -k 10 400 python bench.py --workload c5 --steps 20 --warmup 8 --no-cpu-baseline --layers --layers-out $O/r04_layers_c5.json > /dev/null 2> $O/r04_layers_c5.err
timeout -k 10 400 python bench.py --steps 300 --warmup 40 --no-cpu-baseline > $O/r04_soak_c3.json 2> $O/r04_soak_c3.err
# what the overlapped input pipeline costs / buys: no prefetch at all (voxelise + plan on the training stream, its host sync included),
# two and five arena slots instead of three
TODA_PREFETCH=0 timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r04_bench_c3_noprefetch.json 2> $O/r04_bench_c3_noprefetch.err
TODA_PREFETCH_SLOTS=2 timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r04_bench_c3_slots2.json 2> $O/r04_bench_c3_slots2.err
TODA_PREFETCH_SLOTS=5 timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r04_bench_c3_slots5.json 2> $O/r04_bench_c3_slots5.err
TODA_PREFETCH_ARENA=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_c3_noarena.json 2> $O/r04_bench_c3_noarena.err
unset TODA_BENCH_STEP_MS
fi
if [ "$PART" = "counters" ] || [ "$PART" = "all" ]; then
cd /tmp && export TMPDIR=/tmp
echo "== counters (one pass per set, kernel trace only)"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/r04_pmc_$tag
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r04_pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/r04_pmc_$tag.log 2>&1
  echo "  $tag done"
done
CSVS=$(find /tmp/r04_pmc_FETCH_SIZE /tmp/r04_pmc_WRITE_SIZE /tmp/r04_pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv")
python3 $R/toda_amd/tools/pmc_summary.py gather_gemm_lds_kernel $O/r04_pmc_gather_gemm_64x64.json $CSVS > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py wino_fwd_ws_kernel $O/r04_pmc_wino_fwd.json $CSVS > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py wino_wgrad_kernel $O/r04_pmc_wino_wgrad.json $CSVS > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py wgrad_kernel $O/r04_pmc_sparse_wgrad.json $CSVS > /dev/null
cp $O/r04_pmc_gather_gemm_64x64.json $R/profiles/r04_pmc_gather_gemm_64x64.json
# the C5 workload's dominant kernel (128 -> 128) for that line's roofline.traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/r04_pmc5_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r04_pmc5_$c -- python3 $R/bench.py --workload c5 --steps 2 --warmup 2 --no-cpu-baseline > $O/r04_pmc5_$c.log 2>&1
done
python3 $R/toda_amd/tools/pmc_summary.py "gather_gemm_lds_kernel<8, 8, 1" $O/r04_pmc_gather_gemm_128x128.json $(find /tmp/r04_pmc5_FETCH_SIZE /tmp/r04_pmc5_WRITE_SIZE -name "*counter_collection.csv") > /dev/null
cp $O/r04_pmc_gather_gemm_128x128.json $R/profiles/r04_pmc_gather_gemm_128x128.json
fi
if [ "$PART" = "traces" ] || [ "$PART" = "all" ]; then
cd /tmp && export TMPDIR=/tmp
echo "== where the waves wait (SQ / TA / TCP / TCC counters, one pass per set) and the shader clock (GRBM_GUI_ACTIVE over the dispatch's duration, 8 XCDs)"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TA_BUSY TA_TOTAL_WAVEFRONTS" \
         "TCC_HIT TCC_MISS TCC_REQ TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "GRBM_GUI_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/r04_st_$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/r04_st_$i -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/r04_st_$i.log 2>&1
  echo "  set $i done"
done
ST=$(find /tmp/r04_st_* -name "*counter_collection.csv")
python3 $R/toda_amd/tools/pmc_summary.py gather_gemm_lds_kernel $O/r04_pmc_stall_gather_gemm.json $ST > /dev/null
python3 $R/toda_amd/tools/pmc_summary.py wgrad_kernel $O/r04_pmc_stall_sparse_wgrad.json $ST > /dev/null
fi
if [ "$PART" = "counters" ] || [ "$PART" = "all" ]; then
cd $R
echo "== dense convolutions per layer"
PYTHONPATH=$R timeout -k 10 300 python -m toda_amd.tools.bench_conv2d --config c3 > $O/r04_conv2d_c3.jsonl
PYTHONPATH=$R timeout -k 10 300 python -m toda_amd.tools.bench_conv2d --config c5 > $O/r04_conv2d_c5.jsonl
fi
if [ "$PART" = "traces" ] || [ "$PART" = "all" ]; then
echo "== kernel traces"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r04_kt_c3 /tmp/r04_kt_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_kt_c3 -- python3 $R/bench.py --steps 5 --warmup 4 --no-cpu-baseline > $O/r04_kt_c3.log 2>&1
F=$(find /tmp/r04_kt_c3 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/r04_kt_c3 -name "*kernel_stats.csv" | head -1) $O/r04_bench_c3_kernel_stats.csv
python3 $R/toda_amd/tools/trace_summary.py $F 5 $O/r04_bench_c3_timed_steps.csv > $O/r04_bench_c3_groups.txt
python3 $R/toda_amd/tools/trace_gaps.py $F 5 > $O/r04_gaps_c3.txt
python3 $R/toda_amd/tools/trace_by_shape.py $F 5 $O/r04_sparse_conv_by_launch_shape.csv > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r04_kt_c5 -- python3 $R/bench.py --workload c5 --steps 5 --warmup 4 --no-cpu-baseline > $O/r04_kt_c5.log 2>&1
F5=$(find /tmp/r04_kt_c5 -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_summary.py $F5 5 $O/r04_bench_c5_timed_steps.csv > $O/r04_bench_c5_groups.txt
python3 $R/toda_amd/tools/trace_gaps.py $F5 5 > $O/r04_gaps_c5.txt
# the forward-only workload (BASELINE config 2): steps delimited by the MeanVFE launch
rm -rf /tmp/r04_kt_c2
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r04_kt_c2 -- python3 $R/bench.py --workload c2 --steps 20 --warmup 10 --no-cpu-baseline > $O/r04_kt_c2.log 2>&1
F2=$(find /tmp/r04_kt_c2 -name "*kernel_trace.csv" | head -1)
python3 $R/toda_amd/tools/trace_summary.py $F2 20 $O/r04_bench_c2_timed_steps.csv > $O/r04_bench_c2_groups.txt
python3 $R/toda_amd/tools/trace_gaps.py $F2 20 > $O/r04_gaps_c2.txt
cd $R
# the same C3 line once more behind the nine profiler passes (on two boxes the first runs after rocprofv3 threw a 22-28 ms step
# every fourth or fifth step; both lines are kept)
TODA_BENCH_STEP_MS=1 timeout -k 10 400 python bench.py --steps 50 --warmup 40 --no-cpu-baseline > $O/r04_bench_c3_after_profiler.json 2> $O/r04_bench_c3_after_profiler.err
fi
[ -f $O/r04_bench_c3.json ] && cut -c1-300 $O/r04_bench_c3.json
echo "== done"
