"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: a step is delimited by
consecutive sinkhorn_aggregate_kernel launches (one per pipeline step); the timed steps are the
last `steps` of the pipeline loop (the per-stage block that follows launches the same kernels in
groups of 4, so it is cut off by taking steps from the pipeline-shaped ones only).
usage: step_breakdown.py <trace dir> [first_step last_step]"""
import collections, csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)     # the newest run under that directory
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sinkhorn_aggregate_kernel" in r["Kernel_Name"]]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 12)
agg = collections.defaultdict(lambda: [0, 0.0])
wall = 0.0
for s in range(lo, hi):
    seg = rows[marks[s] : marks[s + 1]]
    wall += (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    for r in seg:
        a = agg[r["Kernel_Name"][:100]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = hi - lo
busy = sum(v[1] for v in agg.values()) / n
print(f"steps {lo}..{hi - 1}: wall {wall / n:.1f} us/step, kernel-busy {busy:.1f} us/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / n:9.1f} us/step  {v[0] / n:6.1f} calls  avg {v[1] / v[0]:8.1f} us  {k}")
