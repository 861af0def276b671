set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for v in 1 0; do
TODA_CLASS_DGRAD=$v timeout -k 10 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --layers --layers-out gpurun_out/l14_$v.json > /dev/null 2> gpurun_out/l14_$v.err
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/l14_1.json'))['kernels']; b=json.load(open('gpurun_out/l14_0.json'))['kernels']
def key(r): return tuple(r['shape'])
B={}
for r in b:
    if r['op']=='gather_gemm': B.setdefault(key(r),[]).append(r)
for r in a:
    if r["op"] in ("gather_gemm","gather_gemm_classed"):
        o=B.get(key(r))
        print(r['op'], r['shape'], r['ms'], r['calls_per_step'], r.get('note'), '| off:', [(x['ms'], x['calls_per_step'], x.get('note')) for x in (o or [])])
PY
