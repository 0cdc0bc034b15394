"""Per-kernel start/end of one pipeline step from a rocprofv3 --kernel-trace CSV: the window around the SALAD kernels with the
idle gaps between consecutive kernels on the same queue.  usage: python scripts/trace_gaps.py <kernel_trace.csv> [pattern]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm256_fuse2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i = idx[len(idx) // 2]                                   # a call from the middle of the run
lo, hi = max(0, i - 6), min(len(rows), i + 8)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = {}
for r in rows[lo:hi]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else float("nan")
    prev_end[q] = e
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"q{q:>3} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  gap-before-on-queue {gap:6.1f}  {name}")
