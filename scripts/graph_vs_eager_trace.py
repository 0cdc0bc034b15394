"""Why is the HIP-graph replay of the retrieval leg slower than the eager launches?
Run under rocprofv3 --kernel-trace, once per mode:
   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/graph_vs_eager_trace.py eager|graph [bf16|fp8]
then  python3 scripts/graph_vs_eager_trace.py analyse OUT_eager OUT_graph
-> per iteration (window between consecutive score-kernel starts): kernels, busy time, idle gaps, the largest gaps."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "knn_scores_kernel" in r["Kernel_Name"]]
    starts = starts[len(starts) // 2:]                     # second half: steady state
    out = []
    for a, b in zip(starts[:-1], starts[1:]):
        w = rows[a:b]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w)
        span = int(rows[b]["Start_Timestamp"]) - int(w[0]["Start_Timestamp"])
        gaps = [(int(w[i + 1]["Start_Timestamp"]) - int(w[i]["End_Timestamp"])) for i in range(len(w) - 1)]
        gaps.append(int(rows[b]["Start_Timestamp"]) - int(w[-1]["End_Timestamp"]))
        names = [r["Kernel_Name"].split("(")[0].split("::")[-1][:28] for r in w]
        out.append((span, busy, gaps, names))
    n = len(out)
    span = sum(o[0] for o in out) / n / 1e3
    busy = sum(o[1] for o in out) / n / 1e3
    print(f"{d}: {n} iterations, {len(out[0][3])} kernels each: period {span:.1f} us, kernels busy {busy:.1f} us, idle {span - busy:.1f} us")
    k = len(out[0][3])
    for j in range(k):
        g = sum(o[2][j] for o in out if len(o[2]) == k) / max(1, sum(1 for o in out if len(o[2]) == k)) / 1e3
        dur = sum(int(0) for o in out)
        print(f"     after {out[0][3][j]:30s} gap {g:6.1f} us")


if sys.argv[1] == "analyse":
    for d in sys.argv[2:]:
        analyse(d)
    sys.exit(0)

import torch
from vpr_amd import ops
from vpr_amd.retrieval import GraphedRetrieval, ShardedGallery
mode, dt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D, B, k = 8448, 64, 10
N = 100_000 if dt == "bf16" else 1_000_000
rows = torch.empty((N, D), dtype=torch.bfloat16 if dt == "bf16" else torch.uint8, device=dev)
scales = torch.empty((N,), dtype=torch.float32, device=dev) if dt == "fp8" else None
for lo in range(0, N, 25000):
    x = torch.nn.functional.normalize(torch.randn(25000, D, device=dev, generator=g), dim=1)
    if dt == "bf16":
        rows[lo:lo + 25000] = x.to(torch.bfloat16)
    else:
        rows[lo:lo + 25000], scales[lo:lo + 25000] = ops.quantize_fp8_rows(x)
gal = ShardedGallery(rows, N, scales=scales)
q = torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1).to(torch.bfloat16)
if mode == "graph":
    gr = GraphedRetrieval(gal, B, k)
    run = lambda: gr(q)
else:
    ws = ops.knn_workspace(B, N, D, k, dev)
    run = lambda: gal.search(q, k, ws)
for _ in range(40):
    run()
torch.cuda.synchronize()
