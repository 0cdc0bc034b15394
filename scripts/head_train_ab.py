"""A/B of vpr_head_train_step's update-kernel geometry (VPR_HEAD_TRAIN_VARIANT: rows per workgroup / batch rows per register
chunk / rows per load group): us per step through vpr_head_train_epoch, per shape."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops

dev = torch.device("cuda:0")
if os.environ.get("VPR_AMD_LIBRARY", "").endswith("ablation.so"):       # timing-only build: parts of the update kernel left out
    ABL = {0: "full", 10: "no stores", 11: "no AdamW arithmetic", 12: "no batch rows / gradient", 13: "no prologue"}
else:
    ABL = None
NAMES = {0: "<8,8,4> (default)", 1: "<8,8,8>", 2: "<4,8,4>", 3: "<8,8,4> nontemporal stores", 4: "<16,8,4>"}
if ABL:
    NAMES = ABL
for (D, hidden, n_out, B) in [(8448, 512, 2, 16), (8448, 512, 2, 64), (8448, 1024, 4, 64)]:
    torch.manual_seed(0)
    N = 4096
    X = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=1)
    Y = torch.randn(N, n_out, device=dev)
    perm = torch.randperm(N, device=dev).to(torch.int32)
    W = [torch.randn(hidden, D, device=dev) * 0.01, torch.zeros(hidden, device=dev), torch.randn(n_out, hidden, device=dev) * 0.04,
         torch.zeros(n_out, device=dev)]
    m, v = ops.head_train_state(W[0], W[2])
    nb = N // B
    res = {}
    for rep in range(2):
        for var in NAMES:
            _lib.tuning_set("VPR_HEAD_TRAIN_VARIANT", var)
            step = 1
            for _ in range(2):
                ops.head_train_epoch(X, Y, perm, B, *W, m, v, step); step += nb
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.head_train_epoch(X, Y, perm, B, *W, m, v, step); step += nb
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(NAMES[var], []).append(round(e0.elapsed_time(e1) * 1e3 / (4 * nb), 2))
    _lib.tuning_set("VPR_HEAD_TRAIN_VARIANT", None)
    print(json.dumps({"shape": f"D={D} hidden={hidden} n_out={n_out} B={B}", "us_per_step": res}), flush=True)
