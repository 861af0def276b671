#!/bin/bash
# Where the Winograd kernels' HBM-side traffic comes from: FETCH_SIZE / WRITE_SIZE per launch of ONE layer shape with
# parts of the kernel switched off in a MEASUREMENT build (-DTODA_ABLATE=1; wrong numbers, never the library that ships):
#   forward / dgrad (TODA_WINO_ABLATE): 0 all on, 1 no input-patch loads, 2 no transformed-filter loads, 16 no output store / hand-off
#   wgrad (TODA_WINO_WG_ABLATE):        0 all on, 1 no x loads, 2 no dy loads
#   make -C toda_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DTODA_ABLATE=1" B=build_ablate OUT=../libtoda_ablate.so
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/scripts/pmc_wino_traffic.sh'
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TODA_HIP_LIB=$R/toda_amd/libtoda_ablate.so
OUT=$O/r05_wino_traffic.txt; : > $OUT
for sh in 0 1; do
  tag=$([ $sh = 0 ] && echo "256->256 @ 2x94x94" || echo "128->128 @ 2x188x188")
  for ab in "0 0" "1 1" "2 2" "16 0"; do
    set -- $ab
    W=""
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/r05_wt_${sh}_$1_$c; rm -rf $d
      WINO_SHAPE=$sh N_IT=6 TODA_WINO_ABLATE=$1 TODA_WINO_WG_ABLATE=$2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/profiles/scripts/wino_bench.py > $O/r05_wt.log 2>&1 || echo "pass failed"
      W="$W $(find $d -name '*counter_collection.csv')"
    done
    python3 - "$tag" "$1" "$2" $W >> $OUT <<'PY'
import csv, sys, collections
tag, ab, abw = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[4:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "wino_fwd_ws_kernel" in k or "wino_wgrad_kernel" in k:
            acc[k.split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    f, w = (sum(c[n]) / max(len(c[n]), 1) for n in ("FETCH_SIZE", "WRITE_SIZE"))
    which = ab if "fwd" in k else abw
    print(f"{tag}  {k:22s} ablate {which:>2s}: fetch {f / 1024:8.1f} MiB  write {w / 1024:8.1f} MiB  ({len(c['FETCH_SIZE'])} launches)")
PY
  done
done
cat $OUT
