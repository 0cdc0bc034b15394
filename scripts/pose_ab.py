"""A/B of the pose head forms and of the split-K factor (VPR_POSE_KS), one process, interleaved: fused single-launch kernel
vs two-launch split form, bench head shape (B = 64, D = 8448, hidden = 1024, n_out = 4) and the reference head (hidden 512)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd import _lib, ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]


for hidden in (1024, 512):
    B, D = 64, 8448
    x = torch.nn.functional.normalize(torch.randn(B, D, device=dev, generator=g), dim=1)
    W1 = torch.randn(hidden, D, device=dev, generator=g) * 0.01
    b1 = torch.zeros(hidden, device=dev)
    W2 = torch.randn(4, hidden, device=dev, generator=g) * 0.05
    b2 = torch.zeros(4, device=dev)
    by = (W1.numel() + W2.numel()) * 4 + x.numel() * 4
    for ks in (None, 4, 8, 12, 16, 24, 32):
        _lib.tuning_set("VPR_POSE_KS", ks)
        row = f"hidden={hidden} KS={ks if ks else 'auto':>4}:"
        for name, kw, var in (("frag+counters", dict(fused=True), 1), ("frag+epilogue", dict(fused=True), 2),
                              ("rowmajor 4 waves + epilogue (default)", dict(fused=False), 0), ("rowmajor 8 waves + epilogue", dict(fused=False), 8)):
            _lib.tuning_set("VPR_POSE_VARIANT", var)
            med, best = timeit(lambda: ops.pose_head(x, W1, b1, W2, b2, 2, **kw))
            row += f"  {name} {med:6.1f} us ({by / med / 1e6 / 8000:.3f})"
        _lib.tuning_set("VPR_POSE_VARIANT", None)
        print(row)
    _lib.tuning_set("VPR_POSE_KS", None)
