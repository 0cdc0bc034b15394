"""What does a fork / join between two HIP streams cost on the launch stream?  Per iteration: main runs two ~40 us GEMMs, a side
stream runs one small kernel that depends on the first GEMM and must finish before the next iteration's first GEMM.
Mechanisms: none (no dependency: the floor), torch events (hipEventRecord + hipStreamWaitEvent), HIP events without the
system-scope fence (light; light_dev adds hipEventReleaseToDevice), stream memory operations
(hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory).  Prints us per iteration."""
import ctypes, os, sys, torch
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpr_amd.backbone import _RawEvents
dev = torch.device("cuda:0")
a = torch.randn(8192, 1024, device=dev, dtype=torch.bfloat16)
w = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
small = torch.randn(64, 1024, device=dev, dtype=torch.bfloat16)
out = torch.empty(8192, 1024, device=dev, dtype=torch.bfloat16)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
sig = [ctypes.c_void_p() for _ in range(2)]
for s in sig:
    assert hip.hipExtMallocWithFlags(ctypes.byref(s), 8, 0x2) == 0
    torch.cuda.synchronize()
counter = [0]
light = {}
for name, flags in (("light", 0x2 | 0x20000000), ("light_dev", 0x2 | 0x40000000), ("nofence_timed", 0x20000000), ("dev_timed", 0x40000000)):
    try:          # hipEventDisableTiming 0x2, hipEventDisableSystemFence 0x20000000, hipEventReleaseToDevice 0x40000000
        light[name] = _RawEvents(dev, flags)
    except RuntimeError as e:
        print(name, e, flush=True)


def run(mode, iters=200):
    def body(i):
        torch.mm(a, w, out=out)
        if mode == "events":
            f = torch.cuda.Event(); f.record(main); side.wait_event(f)
        elif mode in light:
            light[mode].fork(main, side)
        elif mode == "values":
            counter[0] += 1
            assert hip.hipStreamWriteValue32(main.cuda_stream, sig[0], counter[0], 0) == 0
            assert hip.hipStreamWaitValue32(side.cuda_stream, sig[0], counter[0], 0, 0xFFFFFFFF) == 0
        with torch.cuda.stream(side):
            small.mul_(1.0001)
            if mode == "events":
                j = torch.cuda.Event(); j.record(side)
            elif mode in light:
                j = light[mode].mark(side)
            elif mode == "values":
                assert hip.hipStreamWriteValue32(side.cuda_stream, sig[1], counter[0], 0) == 0
        torch.mm(a, w, out=out)
        if mode == "events":
            main.wait_event(j)
        elif mode in light:
            light[mode].join(main, j)
        elif mode == "values":
            assert hip.hipStreamWaitValue32(main.cuda_stream, sig[1], counter[0], 0, 0xFFFFFFFF) == 0
    for i in range(20):
        body(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        body(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for rep in range(2):
    for mode in ["none", "events"] + list(light) + ["values"]:
        print(f"{mode:7s}: {run(mode):7.1f} us per iteration", flush=True)
