"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories.
usage: python scripts/pmc_table.py <dir> [<dir> ...] [--match substr]  ->  JSON {kernel: {counter: avg, "calls": n}}"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short_name

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args:
    args.remove(match)
csv.field_size_limit(sys.maxsize)
acc = {}
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short_name(row["Kernel_Name"])
                if match and match not in k:
                    continue
                e = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                e[0] += float(row["Counter_Value"]); e[1] += 1
out = {k: dict({c: v[0] / v[1] for c, v in cs.items()}, calls=max(v[1] for v in cs.values())) for k, cs in acc.items()}
print(json.dumps(out, indent=1))
