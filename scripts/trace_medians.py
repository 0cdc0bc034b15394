"""Median kernel duration per consecutive run of one kernel in a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev, acc, out = None, [], []
MIN_RUN = int(sys.argv[2]) if len(sys.argv) > 2 else 15
def flush():
    if len(acc) >= MIN_RUN:
        out.append((prev, len(acc), sorted(acc)[len(acc) // 2]))
for r in rows:
    n = r["Kernel_Name"][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n != prev:
        flush(); acc = []; prev = n
    acc.append(d)
flush()
for o in out:
    print("%-72s n=%4d median %8.1f us" % o)
