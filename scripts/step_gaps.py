"""Idle time of the launch queue inside one pipeline step, from a rocprofv3 --kernel-trace CSV of bench.py: for the queue that
runs the score kernel, the gap before every kernel (start - end of the previous kernel on that queue), summed per kernel name.
usage: python scripts/step_gaps.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
PAT = "knn_scores_kernel<false, 208, 2, 4>"     # the score kernel of the pipeline steps (bf16, 100k rows)
knn = [i for i, r in enumerate(rows) if PAT in r["Kernel_Name"]]
q = rows[knn[len(knn) // 2]]["Queue_Id"]
main = [r for r in rows if r["Queue_Id"] == q]
idx = [i for i, r in enumerate(main) if PAT in r["Kernel_Name"]]
idx = [i for i in idx if i + 1 < len(main)]
# steps of the timed region: consecutive score kernels about one step (~11 ms) apart
steps_idx = [j for j in range(1, len(idx)) if 8e6 < int(main[idx[j]]["Start_Timestamp"]) - int(main[idx[j - 1]]["Start_Timestamp"]) < 14e6]
a, b = idx[steps_idx[2] - 1], idx[steps_idx[7]]   # five steady-state steps
span = int(main[b]["Start_Timestamp"]) - int(main[a]["Start_Timestamp"])
busy = 0
gaps = collections.defaultdict(lambda: [0, 0])
for i in range(a + 1, b + 1):
    s, e = int(main[i]["Start_Timestamp"]), int(main[i]["End_Timestamp"])
    pe = int(main[i - 1]["End_Timestamp"])
    busy += e - s
    name = main[i]["Kernel_Name"].split("(")[0][-70:]
    g = max(0, s - pe)
    gaps[name][0] += g; gaps[name][1] += 1
steps = 5
print(f"queue {q}: {span / steps / 1e3:.1f} us per step, kernels busy {busy / steps / 1e3:.1f}, idle {(span - busy) / steps / 1e3:.1f} ({100 * (span - busy) / span:.1f} %), "
      f"{(b - a) / steps:.0f} kernels per step")
for name, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  idle before {name:72s} {g / steps / 1e3:7.1f} us/step  ({n / steps:5.1f} launches, {g / n / 1e3:5.2f} us each)")
