"""DINOv2 ViT backbone in plain PyTorch-ROCm (plumbing: the hand-written HIP path starts at the
token tensor it returns).

Stands in for the backbone half of `torch.hub.load("serizba/salad", "dinov2_salad")`
(dinov2salad/dinov2salad_validation.py:65), which wraps facebookresearch/dinov2 ViT-B/14 and
hands SALAD the final-norm patch tokens + cls token.  Architecture only (random init; no weights
exist offline): patch-14 conv embed, cls token, learned position embedding, pre-norm blocks with
LayerScale, final LayerNorm.  Returns tokens [B, 1+n, C] with the cls token in row 0 — the
layout vpr_salad_aggregate consumes directly (no permute to [B,C,16,16]).
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

def gemm_autotune(enable: bool = True, tuning: bool = True, max_ms_per_gemm: int = 30) -> None:
    """PyTorch TunableOp for the backbone's four hipBLASLt GEMM shapes: during the (untimed) warm-up
    steps each new shape is timed against the library's candidate kernels and the fastest is kept
    in memory (PyTorch also dumps them to a scratch file in the temp dir at exit); call
    gemm_autotune(True, tuning=False) afterwards so the timed region only replays the choices.
    Plumbing around the library GEMMs, not a kernel."""
    import os
    import tempfile
    import torch.cuda.tunable as tun
    tun.enable(enable)
    tun.tuning_enable(enable and tuning)
    if enable and tuning:
        tun.set_max_tuning_duration(max_ms_per_gemm)
        tun.set_filename(os.path.join(tempfile.gettempdir(), f"vpr_tunableop_{os.getpid()}.csv"))


CONFIGS = {
    # name: (embed_dim, depth, heads)
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),       # what the reference's hub entry uses (C = 768)
    "vit_large": (1024, 24, 16),     # BASELINE.json north star (C = 1024)
}


def _ln(norm: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    """LayerNorm: the hand-written HIP kernel for GPU bf16 activations, PyTorch's otherwise (CPU)."""
    if x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048:
        from . import ops
        return ops.layernorm_bf16(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


class _StreamSignals:
    """Fork / join between two HIP streams through stream memory operations (hipStreamWriteValue32 on the producer stream,
    hipStreamWaitValue32 on the consumer stream, one 32-bit word of signal memory per direction, values from a counter that
    only grows) instead of hipEventRecord + hipStreamWaitEvent.  Round-3 experiment, OFF by default: an event operation costs
    the launch stream ~7 us of queue time whether or not anything has to be waited for (scripts/step_gaps.py: 14 us per block
    of the ViT), and in a two-GEMM probe the memory operations are cheaper (scripts/stream_sync_probe.py: 21 -> 13.5 us per
    fork + join pair) — but in the real step they are slower: 11.7-11.9 vs 11.31 ms per step (bench.py --side-sync signals, two
    runs each, one box).  A wait is only ever enqueued AFTER its write has been enqueued, so no stream can be left waiting
    for a value that never comes."""

    def __init__(self, device: torch.device):
        import ctypes
        import os
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))     # the runtime torch itself uses
        hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
        hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        self._hip, self._words, self.count = hip, [], 0
        with torch.cuda.device(device):
            for _ in range(2):
                p = ctypes.c_void_p()
                if hip.hipExtMallocWithFlags(ctypes.byref(p), 8, 0x2) != 0:          # hipMallocSignalMemory
                    raise RuntimeError("hipExtMallocWithFlags(hipMallocSignalMemory) failed")
                self._words.append(p)

    def signal(self, word: int, stream: torch.cuda.Stream, value: int) -> None:
        if self._hip.hipStreamWriteValue32(stream.cuda_stream, self._words[word], value & 0xFFFFFFFF, 0) != 0:
            raise RuntimeError("hipStreamWriteValue32 failed")

    def wait(self, word: int, stream: torch.cuda.Stream, value: int) -> None:
        # hipStreamWaitValueEq: the words only ever hold the value of the latest signal, and a wait for value v is enqueued
        # before the signal for v + 1 can be (program order of the loop), so equality is exact even across the 2^32 wrap
        if self._hip.hipStreamWaitValue32(stream.cuda_stream, self._words[word], value & 0xFFFFFFFF, 0x1, 0xFFFFFFFF) != 0:
            raise RuntimeError("hipStreamWaitValue32 failed")

    # the three-call protocol _blocks_side_chain uses (same as _TorchEvents / _RawEvents)
    def fork(self, main: torch.cuda.Stream, side: torch.cuda.Stream) -> None:
        self.count += 1
        self.signal(0, main, self.count)
        self.wait(0, side, self.count)

    def mark(self, side: torch.cuda.Stream) -> int:
        self.signal(1, side, self.count)
        return self.count

    def join(self, main: torch.cuda.Stream, tick: int) -> None:
        self.wait(1, main, tick)


class _TorchEvents:
    """Fork / join by torch.cuda.Event (hipEventDisableTiming; the record carries a system-scope release)."""

    def fork(self, main: torch.cuda.Stream, side: torch.cuda.Stream) -> None:
        e = torch.cuda.Event()
        e.record(main)
        side.wait_event(e)

    def mark(self, side: torch.cuda.Stream) -> "torch.cuda.Event":
        e = torch.cuda.Event()
        e.record(side)
        return e

    def join(self, main: torch.cuda.Stream, e: "torch.cuda.Event") -> None:
        main.wait_event(e)


class _RawEvents:
    """Fork / join by HIP events created with hipEventDisableTiming | hipEventDisableSystemFence (torch.cuda.Event cannot pass
    the second flag): the record then carries no system-scope cache writeback / invalidate — both streams are queues of
    the same device, device scope is all the hand-over needs.  A ring of events: a wait refers to the record that was
    latest when the wait was enqueued, so re-recording an event later never disturbs a wait already in a queue."""
    RING = 128

    def __init__(self, device: torch.device, flags: int = 0x2 | 0x20000000):
        import ctypes
        import os
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self._hip, self._ring, self._next = hip, [], 0
        with torch.cuda.device(device):
            for _ in range(self.RING):
                e = ctypes.c_void_p()
                st = hip.hipEventCreateWithFlags(ctypes.byref(e), flags)
                if st != 0:
                    raise RuntimeError(f"hipEventCreateWithFlags(0x{flags:x}) failed: hipError {st}")
                self._ring.append(e)

    def __del__(self):
        for e in getattr(self, "_ring", []):
            self._hip.hipEventDestroy(e)

    def _record(self, stream: torch.cuda.Stream):
        e = self._ring[self._next]
        self._next = (self._next + 1) % self.RING
        if self._hip.hipEventRecord(e, stream.cuda_stream) != 0:
            raise RuntimeError("hipEventRecord failed")
        return e

    def _wait(self, stream: torch.cuda.Stream, e) -> None:
        if self._hip.hipStreamWaitEvent(stream.cuda_stream, e, 0) != 0:
            raise RuntimeError("hipStreamWaitEvent failed")

    def fork(self, main: torch.cuda.Stream, side: torch.cuda.Stream) -> None:
        self._wait(side, self._record(main))

    def mark(self, side: torch.cuda.Stream):
        return self._record(side)

    def join(self, main: torch.cuda.Stream, e) -> None:
        self._wait(main, e)


class SplitTokens(NamedTuple):
    """Final-norm tokens as the HIP backbone path holds them: patch [B, n, C] and cls [B, C], both
    contiguous (vpr_salad_aggregate_split consumes the pair without a copy).  token_ready: the backbone's cls-row
    stream already ran `cls_tail_hook` on these cls rows (the SALAD token MLP), ordered before the current stream."""
    patch: torch.Tensor
    cls: torch.Tensor
    token_ready: bool = False

    def joined(self) -> torch.Tensor:
        """[B, 1+n, C], cls first (the reference / torch.hub layout)."""
        return torch.cat([self.cls.unsqueeze(1), self.patch], dim=1)


class Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float = 4.0, init_values: float = 1e-5):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.ls1 = nn.Parameter(init_values * torch.ones(dim))
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)
        self.ls2 = nn.Parameter(init_values * torch.ones(dim))

    folded = False      # inference-only: LayerScale folded into proj / fc2 (saves two passes per block)

    @torch.no_grad()
    def fold_layerscale(self) -> None:
        """ls * (W a + b) == (diag(ls) W) a + ls * b: rewrite proj / fc2 once, then skip the two
        elementwise multiplies in forward.  Changes rounding only (bf16 weights are re-rounded).
        ls1 / ls2 become ones, so the parameters always describe the same function whichever way they
        are read: a state dict saved from a folded model loads into a fresh one correctly (ls = 1 there),
        and `folded` is only the licence to skip a multiply by one."""
        if self.folded:
            return
        for lin, ls in ((self.proj, self.ls1), (self.fc2, self.ls2)):
            w = (ls.float()[:, None] * lin.weight.float()).to(lin.weight.dtype)
            b = (ls.float() * lin.bias.float()).to(lin.bias.dtype)
            lin.weight.copy_(w)
            lin.bias.copy_(b)
            ls.fill_(1.0)
        self.folded = True

    def _load_from_state_dict(self, *args, **kwargs):
        # Any load may bring unfolded weights and real LayerScale values back: the fold licence ends
        # here (the loaded ls1 / ls2 are applied again until fold_layerscale() is called anew).
        super()._load_from_state_dict(*args, **kwargs)
        self.folded = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, C = x.shape
        qkv = self.qkv(_ln(self.norm1, x)).view(B, T, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        y = self.proj(a.transpose(1, 2).reshape(B, T, C))
        x = x + (y if self.folded else self.ls1 * y)
        y = self.fc2(F.gelu(self.fc1(_ln(self.norm2, x))))
        x = x + (y if self.folded else self.ls2 * y)
        return x


class DinoV2(nn.Module):
    def __init__(self, arch: str = "vit_large", img_size: int = 224, patch: int = 14):
        super().__init__()
        dim, depth, heads = CONFIGS[arch]
        self.embed_dim = dim
        self.patch = patch
        self.num_patches = (img_size // patch) ** 2
        self.patch_embed = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + self.num_patches, dim))
        self.blocks = nn.ModuleList(Block(dim, heads) for _ in range(depth))
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        self._register_load_state_dict_pre_hook(self._adopt_foreign_keys)

    # None: 0.1 for facebookresearch/dinov2 (hub) key layouts, 0 for Hugging Face ones (checkpoint.py)
    pos_embed_interpolate_offset: Optional[float] = None

    def _adopt_foreign_keys(self, state_dict, prefix, *unused) -> None:
        """load_state_dict pre-hook (runs for a load through ANY ancestor, e.g. the strict load of
        dinov2salad_validation.py:69 on DINOv2RegressionModel): rewrites, in place, the keys under this
        module's prefix from the hub / serizba-salad (`model.` level, attn.qkv, ls1.gamma, patch_embed.proj,
        mask_token, 37x37 pos_embed) or Hugging Face layout into this class's names — checkpoint.py."""
        from .checkpoint import convert_state_dict
        mine = [k for k in state_dict if k.startswith(prefix)]
        if not mine:
            return
        sub = {}
        for k in mine:
            rest = k[len(prefix):]
            sub[rest[6:] if rest.startswith("model.") else rest] = state_dict.pop(k)
        for k, v in convert_state_dict(sub, self.num_patches, self.pos_embed_interpolate_offset).items():
            state_dict[prefix + k] = v

    @torch.no_grad()
    def forward(self, x: torch.Tensor, split: bool = False):
        """x [B,3,H,W] -> final-norm tokens [B, 1+n, C] (cls first), contiguous; with split=True a
        SplitTokens(patch [B,n,C], cls [B,C]) pair (what the HIP path computes in; no copy)."""
        if self.auto_fold and x.is_cuda and x.dtype == torch.bfloat16 and not all(b.folded for b in self.blocks):
            self.fold_layerscale()      # inference-only module: a load un-folds (Block._load_from_state_dict), the next GPU forward re-folds
        if self._hip_split_ok(x):
            st = self._forward_hip_split(x)
            return st if split else st.joined()
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        if self._hip_ok(x):
            x = self._forward_hip(x)
        else:
            for blk in self.blocks:
                x = blk(x)
            x = _ln(self.norm, x).contiguous()
        return SplitTokens(x[:, 1:].contiguous(), x[:, 0].contiguous()) if split else x

    # GELU of the HIP path's fc1: "tanh" = inside the library GEMM's epilogue (hipBLASLt offers only the tanh
    # form; |gelu_tanh - gelu_erf| <= 4.7e-4, below the bf16 spacing of every output above 0.12) or "erf" = the
    # exact nn.GELU() DINOv2 uses, as a separate in-place pass (+45 us per block at B = 64).  The f32 / CPU block
    # loop is always erf.  evaluate.py (reference checkpoints) selects "erf"; bench.py keeps the default.
    gelu = "tanh"

    def _fc1_gelu(self, blk: "Block", h: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.gelu == "tanh":
            if out is None:
                return torch._addmm_activation(blk.fc1.bias, h, blk.fc1.weight.t(), use_gelu=True)
            return torch._addmm_activation(blk.fc1.bias, h, blk.fc1.weight.t(), use_gelu=True, out=out)
        if self.gelu != "erf":
            raise ValueError("DinoV2.gelu must be 'tanh' or 'erf'")
        hh = torch.addmm(blk.fc1.bias, h, blk.fc1.weight.t()) if out is None else torch.addmm(blk.fc1.bias, h, blk.fc1.weight.t(), out=out)
        return torch._C._nn.gelu_(hh)

    @property
    def _skinny_gelu_mode(self) -> int:
        return 1 if self.gelu == "tanh" else 4

    auto_fold = True        # fold LayerScale at the first bf16 GPU forward after a load, so a freshly loaded checkpoint never
                            # lands on the slow PyTorch block loop by accident (bench.py --no-fold turns it off for the A/B)
    hip_split = True
    cls_after_gemm = True
    fuse_ln_cls = False     # measured: 11.17/11.23 vs 11.22/11.26 ms per step — within noise, so the simpler path is the default

    def _hip_split_ok(self, img: torch.Tensor) -> bool:
        P, C = self.patch, self.embed_dim
        return (self.hip_split and img.is_cuda and img.dtype == torch.bfloat16 and img.dim() == 4
                and img.shape[2] % P == 0 and img.shape[3] % P == 0 and img.shape[3] % 8 == 0
                and (img.shape[2] // P) * (img.shape[3] // P) == self.num_patches
                and img.shape[1] * P * img.shape[3] * 2 <= 48 * 1024
                and C % 8 == 0 and C <= 2048 and C // self.blocks[0].heads == 64 and 1 + self.num_patches <= 288
                and all(b.folded for b in self.blocks))

    def _embed_consts(self, img: torch.Tensor):
        """(patch-embedding weight [C, kpad] bf16, additive token offsets [B*n + B, C] bf16 in the
        split row layout: pos[1+p] + conv bias for the patch rows, cls + pos[0] for the cls rows)."""
        B, Cin = img.shape[0], img.shape[1]
        P, C, n = self.patch, self.embed_dim, self.num_patches
        K = Cin * P * P
        kpad = (K + 63) // 64 * 64
        pe = self.patch_embed
        key = (str(img.device), B, kpad, pe.weight.data_ptr(), pe.weight._version, pe.bias._version,
               self.pos_embed.data_ptr(), self.pos_embed._version, self.cls_token._version)
        # one entry per batch size (an evaluation loop alternates between its full and its last, ragged batch); an entry
        # a HIP graph was captured on is pinned: the graph addresses its tensors for as long as it is replayed
        cache = self.__dict__.setdefault("_embed_cache", {})
        ent = cache.get(key)
        if ent is None:
            for k in [k for k, e in cache.items() if not e[2]][: max(0, len(cache) - 3)]:
                del cache[k]
            dev = img.device
            w = torch.zeros((C, kpad), dtype=torch.bfloat16, device=dev)
            w[:, :K] = pe.weight.detach().reshape(C, K).to(dev, torch.bfloat16)
            pos = self.pos_embed.detach().float().to(dev)[0]                             # [1+n, C]
            body = (pos[1:] + pe.bias.detach().float().to(dev)).to(torch.bfloat16)       # [n, C]
            tail = (pos[0] + self.cls_token.detach().float().to(dev).view(C)).to(torch.bfloat16)
            off = torch.cat([body.unsqueeze(0).expand(B, -1, -1).reshape(B * n, C),
                             tail.unsqueeze(0).expand(B, -1)], dim=0).contiguous()
            ent = cache[key] = [w, off, False]
        if not ent[2] and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            ent[2] = True
        return ent[0], ent[1]

    cls_side_chain = True
    side_sync = "events"        # fork / join of the cls side chain: "events" (torch.cuda.Event), "light" (HIP events without the system-scope
                                # fence, _RawEvents), "signals" (stream memory operations, _StreamSignals: measured SLOWER in the step,
                                # 11.7-11.9 vs 11.31 ms, although the two-GEMM probe favours them).  bench.py --side-sync

    def _blocks_side_chain(self, x: torch.Tensor, h: torch.Tensor, cum: torch.Tensor, B: int, n: int, C: int) -> "SplitTokens":
        """The 24 blocks with the cls rows on their own stream.  Between two attentions the B cls rows need six
        small kernels (proj, LayerNorm, fc1, fc2, LayerNorm, next qkv: ~40 us) that depend on nothing but the cls
        rows of the attention output; in-stream they cost 0.8 ms per step, almost all of it launch latency.  Here
        the side stream is forked right after each attention and joined right before the next one, ~375 us of
        patch-row GEMMs later, so the join never waits (an earlier attempt forked and joined around every single
        cls-row launch and lost 0.6 ms to cross-queue waits).  The main stream touches only patch rows
        (x[:Mp], LayerNorm on the patch slice), the side stream only cls rows: no shared writes.
        Lifetimes: `att` (read by the side stream) and `qkv_next` (written by it) stay referenced until after
        the join; side-stream temporaries come from that stream's allocator pool."""
        from . import ops
        Mp, M = B * n, B * n + B
        dev, bf = x.device, torch.bfloat16
        blocks = self.blocks
        main = torch.cuda.current_stream(dev)
        sides = self.__dict__.setdefault("_sides", {})      # one side stream per (device, main stream)
        skey = (str(dev), main.cuda_stream)
        side = sides.get(skey)
        if side is None:
            # high priority: the cls-row workgroups take the first CU slots a retiring GEMM workgroup frees (the
            # library GEMMs fill every CU, so at equal priority the small kernels sit in the queue: 40-95 us each)
            import os
            prio = int(os.environ.get("VPR_SIDE_PRIORITY", "-1"))
            side = sides[skey] = torch.cuda.Stream(device=dev, priority=prio)
        mode = self.side_sync if not torch.cuda.is_current_stream_capturing() else "events"     # capture: plain event nodes
        if mode not in ("events", "light", "signals"):
            raise RuntimeError(f"DinoV2.side_sync: unknown mode {mode!r}")
        if mode == "events":
            sync = _TorchEvents()
        else:
            syncs = self.__dict__.setdefault("_syncs", {})
            sync = syncs.get(skey + (mode,))
            if sync is None:
                sync = syncs[skey + (mode,)] = _StreamSignals(dev) if mode == "signals" else _RawEvents(dev)
        C3, C4 = blocks[0].qkv.weight.shape[0], blocks[0].fc1.weight.shape[0]
        xp, xc = x[:Mp], x[Mp:]
        hp = h[:Mp]
        qkv = torch.empty((M, C3), dtype=bf, device=dev)
        torch.addmm(blocks[0].qkv.bias, hp, blocks[0].qkv.weight.t(), out=qkv[:Mp])
        ops.skinny_linear_bf16(h[Mp:], blocks[0].qkv.weight, blocks[0].qkv.bias, qkv[Mp:], 0)
        cls_out, hooked = None, False
        for i, blk in enumerate(blocks):
            last = i + 1 == len(blocks)
            nb = blocks[i + 1] if not last else None
            nxt = nb.norm1 if not last else self.norm
            att = ops.attention_qkv_split_bf16(qkv, B, 1 + n, n, blk.heads)
            qkv_next = torch.empty((M, C3), dtype=bf, device=dev) if not last else None
            sync.fork(main, side)
            with torch.cuda.stream(side):
                ops.skinny_linear_bf16(att[Mp:], blk.proj.weight, None, xc, 2)
                hc = ops.bias_layernorm_bf16(xc, cum[2 * i], blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
                hhc = torch.empty((B, C4), dtype=bf, device=dev)
                ops.skinny_linear_bf16(hc, blk.fc1.weight, blk.fc1.bias, hhc, self._skinny_gelu_mode)
                ops.skinny_linear_bf16(hhc, blk.fc2.weight, None, xc, 2)
                hc2 = ops.bias_layernorm_bf16(xc, cum[2 * i + 1], nxt.weight, nxt.bias, nxt.eps)
                if not last:
                    ops.skinny_linear_bf16(hc2, nb.qkv.weight, nb.qkv.bias, qkv_next[Mp:], 0)
                else:
                    cls_out = hc2
                    if self.cls_tail_hook is not None:     # e.g. SALAD's token MLP: needs the cls rows only, rides on this stream
                        self.cls_tail_hook(cls_out, main.cuda_stream)
                        hooked = True
                joined = sync.mark(side)
            xp.addmm_(att[:Mp], blk.proj.weight.t())
            hp = ops.bias_layernorm_bf16(xp, cum[2 * i], blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            hh = self._fc1_gelu(blk, hp)
            xp.addmm_(hh, blk.fc2.weight.t())
            hp = ops.bias_layernorm_bf16(xp, cum[2 * i + 1], nxt.weight, nxt.bias, nxt.eps)
            if not last:
                torch.addmm(nb.qkv.bias, hp, nb.qkv.weight.t(), out=qkv_next[:Mp])
            sync.join(main, joined)        # the side chain finished ~0.3 ms ago: satisfied on arrival
            qkv = qkv_next                 # (att stayed referenced up to here)
        return SplitTokens(hp.view(B, n, C), cls_out, hooked)

    # optional callable(cls_rows [B, C] bf16, raw handle of the main stream): run on the cls-row side stream right after the
    # final LayerNorm of the cls rows, before that stream is joined (modules.DinoV2Salad installs SALAD's token MLP here)
    cls_tail_hook = None

    def _raw_tokens(self, M: int, Mp: int, C: int, dev: torch.device) -> torch.Tensor:
        bufs = self.__dict__.setdefault("_raw_bufs", {})     # per (device, stream, shape): two streams never share it
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream, M, C)
        ent = bufs.get(key)
        if ent is None:
            if len(bufs) > 8:
                for k in [k for k, e in bufs.items() if not e[1]]:       # entries a HIP graph addresses stay
                    del bufs[k]
            ent = bufs[key] = [torch.zeros((M, C), dtype=torch.bfloat16, device=dev), False]
        if not ent[1] and torch.cuda.is_current_stream_capturing():
            ent[1] = True
        return ent[0]

    def _forward_hip_split(self, img: torch.Tensor) -> "SplitTokens":
        """The whole backbone on the GPU in the split row layout [B*n patch rows | B cls rows]:
        every linear layer runs as one library GEMM over the B*n patch rows — at n = 256 an exact
        number of 256-row tiles (64 tile rows at B = 64; the cls-first [B, 257, C] layout gives 64.25,
        i.e. a fourth, almost empty wave of tiles: hipBLASLt measures 88/117/96 us instead of
        105/133/148 us for qkv/fc1/fc2) — plus a 64-row GEMM over the cls rows.
        * patch embedding = HIP patchify + one GEMM (instead of MIOpen's implicit-GEMM conv +
          transposes + cat + add); cls / position / conv-bias terms are a static additive matrix
          consumed by the first LayerNorm kernel, which writes the residual stream anyway;
        * the residual add lives in the proj / fc2 GEMM (`x.addmm_`, beta = 1, in place) and their
          biases never enter the bf16 stream: their running sum is a static [C] f32 vector per
          LayerNorm, added inside the kernel (vpr_bias_layernorm_bf16);
        * bias + GELU in the fc1 GEMM epilogue (hipBLASLt's tanh form; vs an f32 reference the max
          error equals erf-GELU's 0.016: bf16 rounding dominates);
        * attention: vpr_attention_qkv_split_bf16 (token -> row mapping inside the kernel)."""
        from . import ops
        B, n, C, P = img.shape[0], self.num_patches, self.embed_dim, self.patch
        Mp, M = B * n, B * n + B
        dev, bf = img.device, torch.bfloat16
        w, off = self._embed_consts(img)
        a = ops.patchify_bf16(img.contiguous(), P, w.shape[1], 0)               # [Mp, kpad]
        raw = self._raw_tokens(M, Mp, C, dev)       # persistent: its cls rows are zero and nothing ever writes them
        torch.mm(a, w.t(), out=raw[:Mp])
        blocks = self.blocks
        n0 = blocks[0].norm1
        x, h = ops.add_layernorm_bf16(raw, off, n0.weight, n0.bias, n0.eps)
        cum = self._cumulative_bias(dev)

        if self.cls_side_chain:
            return self._blocks_side_chain(x, h, cum, B, n, C)

        # patch rows: library GEMMs; cls rows: vpr_skinny_linear_bf16 (a library GEMM spends 9-14 us on 64 rows).
        # cls_after_gemm: each cls-row launch comes right AFTER the library GEMM that used the same weight
        # matrix (the 2-8 MB it streams could then be cache-resident).  Measured equal to "before":
        # 11.20 / 11.16 vs 11.12 / 11.17 ms per step — the launches are latency-, not bandwidth-bound.
        # Tried and dropped: proj over all B*n + B rows in one GEMM (50 us vs 38 + 8); the cls-row kernels on
        # a side stream (cross-queue waits: 12.8 vs 12.2 ms/step).
        # fuse_ln_cls (option): the two cls-row linears that directly follow a LayerNorm (qkv, fc1) ride in
        # the LayerNorm's launch (vpr_bias_layernorm_cls_linear_bf16): 0.03-0.05 ms per step, default off.
        fuse = self.fuse_ln_cls and C % 32 == 0 and self.gelu == "tanh"     # the fused launch has the tanh form only
        after = self.cls_after_gemm
        rs = torch.empty((C // 16, B, 2), dtype=torch.float32, device=dev) if fuse else None   # stream-ordered reuse
        fc = self._cls_fused_consts(dev) if fuse else None
        C3, C4 = blocks[0].qkv.weight.shape[0], blocks[0].fc1.weight.shape[0]

        def pair(big, small, do_small=True):     # the library GEMM on the patch rows and the cls-row launch, in either order
            if do_small and not after:
                small()
            big()
            if do_small and after:
                small()

        qkv = torch.empty((M, C3), dtype=bf, device=dev)
        cls_qkv_pending = True                   # False once a fused LayerNorm launch has produced qkv[Mp:]
        for i, blk in enumerate(blocks):
            last = i + 1 == len(blocks)
            pair(lambda: torch.addmm(blk.qkv.bias, h[:Mp], blk.qkv.weight.t(), out=qkv[:Mp]),
                 lambda: ops.skinny_linear_bf16(h[Mp:], blk.qkv.weight, blk.qkv.bias, qkv[Mp:], 0), cls_qkv_pending)
            att = ops.attention_qkv_split_bf16(qkv, B, 1 + n, n, blk.heads)
            pair(lambda: x[:Mp].addmm_(att[:Mp], blk.proj.weight.t()),
                 lambda: ops.skinny_linear_bf16(att[Mp:], blk.proj.weight, None, x[Mp:], 2, cum[2 * i] if fuse else None, rs))
            hh = torch.empty((M, C4), dtype=bf, device=dev)
            n2 = blk.norm2
            if fuse:
                h = ops.bias_layernorm_cls_linear_bf16(x, cum[2 * i], n2.weight, n2.bias, n2.eps, Mp, rs,
                                                       fc[2 * i + 1], hh[Mp:], gelu=True)
            else:
                h = ops.bias_layernorm_bf16(x, cum[2 * i], n2.weight, n2.bias, n2.eps)
            pair(lambda: self._fc1_gelu(blk, h[:Mp], out=hh[:Mp]),
                 lambda: ops.skinny_linear_bf16(h[Mp:], blk.fc1.weight, blk.fc1.bias, hh[Mp:], self._skinny_gelu_mode), not fuse)
            pair(lambda: x[:Mp].addmm_(hh[:Mp], blk.fc2.weight.t()),
                 lambda: ops.skinny_linear_bf16(hh[Mp:], blk.fc2.weight, None, x[Mp:], 2,
                                                cum[2 * i + 1] if fuse and not last else None, rs if not last else None))
            if not last:
                nb = blocks[i + 1]
                qkv = torch.empty((M, C3), dtype=bf, device=dev)
                if fuse:
                    h = ops.bias_layernorm_cls_linear_bf16(x, cum[2 * i + 1], nb.norm1.weight, nb.norm1.bias, nb.norm1.eps,
                                                           Mp, rs, fc[2 * i + 2], qkv[Mp:])
                else:
                    h = ops.bias_layernorm_bf16(x, cum[2 * i + 1], nb.norm1.weight, nb.norm1.bias, nb.norm1.eps)
                cls_qkv_pending = not fuse
            else:
                h = ops.bias_layernorm_bf16(x, cum[2 * i + 1], self.norm.weight, self.norm.bias, self.norm.eps)
        return SplitTokens(h[:Mp].view(B, n, C), h[Mp:])

    gelu_in_epilogue = True

    def _hip_ok(self, x: torch.Tensor) -> bool:
        return (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048
                and all(b.folded for b in self.blocks))

    residual_in_gemm = True

    def _forward_hip(self, x: torch.Tensor) -> torch.Tensor:
        """Same math as the block loop, with x carrying the residual stream and h the normalised copy.
        residual_in_gemm: the residual add lives in the proj / fc2 GEMM (`x.addmm_(a, W^T)`, beta = 1,
        in place: the f32 accumulator is added to the stream before the single bf16 rounding) and
        their biases are not written into the stream at all: the running sum of the biases is a
        static [C] f32 vector per LayerNorm, added inside the kernel (vpr_bias_layernorm_bf16).
        Per block that is LN r1 w1 twice (67 MB each) instead of add+LN r2 w2 (135 MB each), for
        ~8 us of extra C reads in the two GEMMs.  Otherwise: every residual add fused into the
        LayerNorm that follows it (vpr_add_layernorm_bf16)."""
        from . import ops
        B, T, C = x.shape
        x = x.contiguous()
        blocks = self.blocks
        n0 = blocks[0].norm1
        h = ops.layernorm_bf16(x, n0.weight, n0.bias, n0.eps)
        hip_attn = (C // blocks[0].heads == 64) and T <= 288      # the short-sequence HIP kernel's domain
        if self.residual_in_gemm:
            cum = self._cumulative_bias(x.device)
            x2 = x.view(B * T, C)          # fresh tensor from forward(): safe to update in place
        for i, blk in enumerate(blocks):
            if hip_attn:
                a = ops.attention_qkv_bf16(blk.qkv(h), blk.heads)
            else:
                qkv = blk.qkv(h).view(B, T, 3, blk.heads, C // blk.heads).permute(2, 0, 3, 1, 4)
                a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, T, C)
            nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else self.norm
            if self.residual_in_gemm:
                x2.addmm_(a.view(B * T, C), blk.proj.weight.t())
                h = ops.bias_layernorm_bf16(x2, cum[2 * i], blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
                hh = self._fc1_gelu(blk, h)
                x2.addmm_(hh, blk.fc2.weight.t())
                h = ops.bias_layernorm_bf16(x2, cum[2 * i + 1], nxt.weight, nxt.bias, nxt.eps).view(B, T, C)
                continue
            y = blk.proj(a)
            x, h = ops.add_layernorm_bf16(x, y, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            if self.gelu_in_epilogue:
                # bias + GELU inside the fc1 GEMM epilogue (hipBLASLt's tanh form): removes a 270 MB
                # elementwise pass per block.  Its deviation from erf-GELU (<= 3e-4) is below bf16
                # resolution: against an f32 reference both forms measure the same max error (0.016).
                hh = self._fc1_gelu(blk, h.view(B * T, C))
                y = blk.fc2(hh).view(B, T, C)
            else:
                y = blk.fc2(F.gelu(blk.fc1(h)))
            x, h = ops.add_layernorm_bf16(x, y, nxt.weight, nxt.bias, nxt.eps)
        return h.view(B, T, C)

    def _cls_fused_consts(self, device: torch.device):
        """ClsLinearConsts of the LayerNorm-fused cls-row linears: entry 2i = block i's (norm1, qkv) with the
        cumulative bias before it (entry 0 unused: block 0's first LayerNorm is the embedding add+LN), entry
        2i+1 = block i's (norm2, fc1).  Static, rebuilt when the parameters change identity / version."""
        from . import ops
        ps = [p for b in self.blocks for p in (b.qkv.weight, b.qkv.bias, b.fc1.weight, b.fc1.bias, b.proj.bias, b.fc2.bias,
                                               b.norm1.weight, b.norm1.bias, b.norm2.weight, b.norm2.bias)]
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_clsf_key", None) != key:
            cum = self._cumulative_bias(device)
            out = []
            for i, b in enumerate(self.blocks):
                out.append(None if i == 0 else
                           ops.ClsLinearConsts.build(b.qkv.weight, b.qkv.bias, b.norm1.weight, b.norm1.bias, cum[2 * i - 1]))
                out.append(ops.ClsLinearConsts.build(b.fc1.weight, b.fc1.bias, b.norm2.weight, b.norm2.bias, cum[2 * i]))
            self._clsf, self._clsf_key = out, key
        return self._clsf

    def _cumulative_bias(self, device: torch.device):
        """cum[2i] = sum of proj/fc2 biases up to and including block i's proj; cum[2i+1] adds its fc2
        (f32, one [2L, C] tensor; rebuilt when the biases change identity, e.g. after a state-dict load)."""
        key = (str(device),) + tuple(b.proj.bias.data_ptr() for b in self.blocks) + tuple(b.proj.bias._version for b in self.blocks)
        if getattr(self, "_cum_key", None) != key:
            rows, acc = [], torch.zeros(self.embed_dim, dtype=torch.float32, device=device)
            for blk in self.blocks:
                acc = acc + blk.proj.bias.detach().float().to(device)
                rows.append(acc)
                acc = acc + blk.fc2.bias.detach().float().to(device)
                rows.append(acc)
            self._cum_bias, self._cum_key = torch.stack(rows).contiguous(), key
        return self._cum_bias

    def fold_layerscale(self) -> "DinoV2":
        for blk in self.blocks:
            blk.fold_layerscale()
        return self

    def flops_per_image(self) -> float:
        T, C, L = 1 + self.num_patches, self.embed_dim, len(self.blocks)
        per_block = 2 * T * C * 3 * C + 2 * T * C * C + 4 * T * T * C + 2 * 2 * T * C * 4 * C
        return L * per_block + 2 * self.num_patches * C * 3 * self.patch * self.patch
