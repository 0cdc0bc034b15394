import os, sys, torch, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vpr_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C = 64, 1024
patch = torch.randn(B, 256, C, device=dev, generator=g).to(torch.bfloat16)
cls = torch.randn(B, C, device=dev, generator=g).to(torch.bfloat16)
r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02)
w = ops.SaladWeights(w1_sc=r(1024, C).bfloat16(), b1_sc=r(1024), w2_s=r(64, 512).bfloat16(), b2_s=r(64),
                     w2_c=r(128, 512).bfloat16(), b2_c=r(128), w1_t=r(512, C).bfloat16(), b1_t=r(512),
                     w2_t=r(256, 512).bfloat16(), b2_t=r(256), dustbin=1.0)
def timeit(fn, iters=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return round(ts[len(ts) // 2] * 1e3, 1)
res, outs = {}, {}
for rep in range(3):
    for var in (0, 4):
        _lib.tuning_set("VPR_SALAD_VARIANT", var)
        res.setdefault(var, []).append(timeit(lambda: ops.salad_aggregate_split(patch, cls, w, 3, True)))
        outs[var] = ops.salad_aggregate_split(patch, cls, w, 3, True)[0].clone()
_lib.tuning_set("VPR_SALAD_VARIANT", None)
print(json.dumps({"us_onecall": res, "bit_identical": bool(torch.equal(outs[0], outs[4]))}))
