"""per-queue view of one timed step of a bench.py kernel trace: which non-toda kernels run on which queue, and in which neighbourhood"""
import csv, sys, collections, re
sys.path.insert(0, sys.argv[3] if len(sys.argv) > 3 else '/root/repo')
from toda_amd.tools.trace_summary import timed_window
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2])
t0, t1 = timed_window(rows, steps)
sel = [r for r in rows if t0 < int(r["End_Timestamp"]) <= t1]
qk = "Queue_Id" if "Queue_Id" in sel[0] else "Stream_Id"
print("columns:", list(sel[0].keys()))
def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    if "toda::" in n: n = "T:" + n.split("toda::")[1].split("(")[0]
    return n[:90]
byq = collections.defaultdict(list)
for r in sel: byq[r[qk]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6 / steps
    nt = [r for r in rs if "toda::" not in r["Kernel_Name"]]
    print(f"== queue {q}: {len(rs)/steps:.1f} launches/step, {tot:.3f} ms/step busy; non-toda {len(nt)/steps:.1f}/step {sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in nt)/1e6/steps:.3f} ms/step")
# the main queue = most busy: print one step's sequence compactly (runs of toda kernels collapsed)
import os
order = sorted(byq.items(), key=lambda kv: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))
main = order[int(os.environ.get("TS_QUEUE_RANK", "0"))][1]
span = (t1 - t0) // steps
one = [r for r in main if int(r["End_Timestamp"]) > t1 - span]
run = []
for r in one:
    n = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith("T:"):
        run.append(n[2:].split("<")[0])
    else:
        if run:
            print("   [" + " ".join(run[-3:]) + f"] ({len(run)} toda)"); run = []
        print(f"  {d:7.1f} us  {n}")
if run: print("   [" + " ".join(run[-3:]) + f"] ({len(run)} toda)")
