"""Replay a model forward from a HIP graph, one graph per input shape.

The evaluation / gallery-building loops are host-bound once the GPU path is in place: a ViT-L/14 + SALAD forward is
~300 kernel launches = 4-6 ms of Python under the GIL per batch, the same lock the image-decode threads
(`loader.ImageBatchLoader`) need.  Captured once per input shape, the forward is ONE graph launch (0.1 ms of host time,
same GPU time as the eager in-stream forward: `scripts/graph_backbone.py`).

Capture rules the wrapped function must obey (the HIP paths of this package do): no host sync, no `.item()`/`.cpu()`,
workspaces from `ops.workspace`.  Those caches are keyed by stream, so the warm-up calls and the capture run on the SAME
private stream: the buffers the warm-up allocated are the ones the graph addresses (nothing is allocated a second time
inside the capture), and `close()` — or garbage collection of the wrapper — drops that stream's cache entries again.  The DINOv2 backbone's cls-row
side stream is switched off inside the captured forward: fork / join branches replay slower than the eager side stream
(16.2 vs 10.9 ms) while the linear chain replays at the eager in-stream time (11.3 ms); results are bit-identical
either way (`test_cls_side_chain_is_bit_identical_to_in_stream_path`, `test_graphed_forward_equals_eager`).

A captured graph addresses its operands by raw pointer, including what the package caches between calls (workspaces,
packed weights, batch-size-keyed backbone constants).  Those caches pin every entry that is handed out while a capture
is in progress (`ops.capturing()`): a pinned entry is never freed or replaced in place, so graphs of several input
shapes and eager calls can be interleaved freely.

Weights are addressed the same way: after changing parameters (a state-dict load, `fold_layerscale()`, `pack()`), build
a new GraphedForward.

The returned tensors are the graph's static output buffers: valid until the next call with the same input shape —
consume (or clone) them before that, on the stream the call was made on.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch


def _backbones(obj) -> list:
    """DinoV2 backbones reachable from a module (their cls side chain is disabled while a graph is captured)."""
    out = []
    if isinstance(obj, torch.nn.Module):
        for m in obj.modules():
            if hasattr(m, "cls_side_chain"):
                out.append(m)
    return out


class GraphedForward:
    def __init__(self, fn: Callable[[torch.Tensor], object], module: torch.nn.Module = None, warmup: int = 2):
        """fn(x) -> tensor or (nested) tuple of tensors.  `module` (optional): where to look for DINOv2 backbones whose
        side stream has to be off during capture (defaults to fn if it is a module)."""
        self.fn, self.warmup = fn, max(1, warmup)
        self._backbones = _backbones(module if module is not None else fn)
        self._graphs: Dict[Tuple, Tuple] = {}
        self._streams: Dict[int, torch.cuda.Stream] = {}      # device index -> the private warm-up / capture stream
        self.fallback_reason = None          # set when a capture failed: the wrapper then runs the eager forward for good

    def close(self) -> None:
        """Destroy the graphs and release what the package's stream-keyed caches hold for this wrapper's private
        streams (workspaces, the backbone's raw-token buffers)."""
        from . import ops
        self._graphs.clear()
        for s in self._streams.values():
            ops.drop_stream_caches(s.cuda_stream)
            for b in self._backbones:
                bufs = b.__dict__.get("_raw_bufs", {})
                for k in [k for k in bufs if k[1] == s.cuda_stream]:
                    del bufs[k]
        self._streams.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:                     # noqa: BLE001  interpreter shutdown
            pass

    def _run(self, x):
        saved = [b.cls_side_chain for b in self._backbones]
        for b in self._backbones:
            b.cls_side_chain = False
        try:
            with torch.no_grad():
                return self.fn(x)
        finally:
            for b, s in zip(self._backbones, saved):
                b.cls_side_chain = s

    def __call__(self, x: torch.Tensor):
        if not x.is_cuda or self.fallback_reason is not None:
            return self._run(x)
        key = (tuple(x.shape), x.dtype, x.device.index)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = x.clone()
            cur = torch.cuda.current_stream(x.device)
            side = self._streams.get(x.device.index)
            if side is None:
                side = self._streams[x.device.index] = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):                      # warm-up outside capture: workspaces, module load, plane caches
                for _ in range(self.warmup):
                    self._run(static_in)
            cur.wait_stream(side)
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, stream=side):    # the warm-up's stream: its stream-keyed buffers are re-used
                    static_out = self._run(static_in)
            except Exception as e:                             # noqa: BLE001  a forward that cannot be captured (host sync, ...)
                self.fallback_reason = f"{type(e).__name__}: {e}"
                torch.cuda.synchronize(x.device)
                return self._run(x)
            entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(x)
        graph.replay()
        return static_out

    def graphs(self) -> int:
        return len(self._graphs)
