"""End-to-end throughput of the VPR + geopose hot path on N MI355X GPUs (one process per GPU).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher in the environment: the script starts the N ranks itself, same command as above)

A step = one batch of B synthetic 3x224x224 images per GPU through DINOv2 ViT-L/14 (PyTorch-ROCm,
random init) -> SALAD aggregation (HIP) -> bf16 cosine top-k against the 100k-row synthetic
gallery, row-sharded over the ranks (HIP; RCCL all-gather of queries and of per-shard top-k,
on-device merge) -> fused (lat, lon, sin, cos) head (HIP).  Weak scaling: B per GPU is fixed.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
hand-written kernel (the HBM-bound kNN score kernel, timed live with HIP events inside the timed
region), `kernels` (one roofline row per other hand-written stage; the SALAD row is measured live
inside the timed region too), `recall_at_1` (planted positives, through all-gather + merge when the
gallery is sharded), `dist` (process group, ranks it really has, exchange time per step) and
`cpu_baseline` (oracle/ + the same backbone on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory instead of host-coherent memory: ~1 us less dispatch latency per kernel on this
# runtime, ~300 kernels per step: 11.74 -> 11.41 ms per step (same box, twice).  Read by the HIP runtime when it
# initialises, so it has to be in the environment before the first HIP call; an explicit setting wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# dmabuf IPC (the only form this pool's driver supports): RCCL across processes fails with hipIpcGetMemHandle without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D_DESC = 8448
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense fp8 (block-scaled f8f6f4 MFMA), same guide
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class _StdoutToStderr:
    """File descriptor 1 -> 2 for the duration of the block.  RCCL 2.26 prints a version banner ("RCCL version : ...",
    five lines) to STDOUT when its first communicator is created; this run's stdout is reserved for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def make_gallery_shard(rows: int, seed: int, dev) -> torch.Tensor:
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty((rows, D_DESC), dtype=torch.bfloat16, device=dev)
    step = 16384
    for lo in range(0, rows, step):
        n = min(step, rows - lo)
        out[lo:lo + n] = torch.nn.functional.normalize(torch.randn(n, D_DESC, device=dev, generator=g), dim=1).to(torch.bfloat16)
    return out


def cpu_baseline(ext_state, arch, head_mods, images_cpu, gallery_sample_cpu, n_total, k):
    """Same op sequence on the host: backbone (PyTorch CPU fp32) + oracle SALAD / kNN / head.
    Bounded sample: `images_cpu` (a few images) and a slice of the gallery, scaled to n_total."""
    from oracle import heads as oheads, salad as osalad
    from vpr_amd.modules import DinoV2Salad
    threads = min(os.cpu_count(), 16)          # the GPU box gives one GPU's job a 16-core share
    torch.set_num_threads(threads)
    ext = DinoV2Salad(arch).float().eval()
    ext.load_state_dict(ext_state)
    b = images_cpu.shape[0]
    t0 = time.perf_counter()
    with torch.no_grad():
        tokens = ext.backbone(images_cpu.float())
    t_backbone = time.perf_counter() - t0
    agg = ext.aggregator
    m2 = lambda w: w.detach().reshape(w.shape[0], -1).float()
    w = dict(w1_sc=torch.cat([m2(agg.score[0].weight), m2(agg.cluster_features[0].weight)], 0),
             b1_sc=torch.cat([agg.score[0].bias, agg.cluster_features[0].bias], 0).detach().float(),
             w2_s=m2(agg.score[3].weight), b2_s=agg.score[3].bias.detach().float(),
             w2_c=m2(agg.cluster_features[3].weight), b2_c=agg.cluster_features[3].bias.detach().float(),
             w1_t=m2(agg.token_features[0].weight), b1_t=agg.token_features[0].bias.detach().float(),
             w2_t=m2(agg.token_features[2].weight), b2_t=agg.token_features[2].bias.detach().float())
    t0 = time.perf_counter()
    desc = osalad.salad_aggregate(tokens, w, float(agg.dust_bin), 3, dtype=torch.float32, quantize=False)
    t_salad = time.perf_counter() - t0
    q = desc.to(torch.bfloat16).float()
    gal32 = gallery_sample_cpu.float()                    # a CPU deployment would hold the gallery in fp32
    t0 = time.perf_counter()
    s = q @ gal32.T                                       # fp32 brute force, all threads
    torch.topk(s, k, dim=1)
    t_knn = (time.perf_counter() - t0) * (n_total / gallery_sample_cpu.shape[0])
    W1, b1, W2, b2 = head_mods
    t0 = time.perf_counter()
    oheads.mlp_head(desc, W1, b1, W2, b2, 2, dtype=torch.float32)
    t_head = time.perf_counter() - t0
    total = t_backbone + t_salad + t_knn + t_head
    return {"value": b / total, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{b} images (backbone {t_backbone:.2f}s, SALAD {t_salad:.2f}s, head {t_head:.3f}s) + "
                      f"fp32 brute-force kNN on {gallery_sample_cpu.shape[0]} of {n_total} gallery rows scaled x"
                      f"{n_total / gallery_sample_cpu.shape[0]:.0f} ({t_knn:.2f}s); torch {torch.__version__} CPU, "
                      f"{threads} threads"}


def _avg_ms(fn, n=10, warm=2):
    """Average device time of fn() over n calls, HIP events on the current stream (the one the kernels launch on)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pmc_traffic(summary: str, kernel: str):
    """HBM bytes per launch of `kernel` from a committed PMC summary (scripts/pmc_summary.py), only if that summary was
    measured on the kernel source this run is about to launch (hash of knn.hip + vpr_common.h) -> (bytes, source) or (None, None)."""
    path = os.path.join(ROOT, "profiles", summary)
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from pmc_summary import kernel_source_sha16
        with open(path) as f:
            pj = json.load(f)
        if pj.get("source_sha16") == kernel_source_sha16(ROOT) and kernel in pj.get("kernels", {}):
            return pj["kernels"][kernel]["hbm_bytes_per_launch"], f"profiles/{summary} (source_sha16 {pj['source_sha16']}, {kernel})"
    except Exception:                                           # noqa: BLE001  a missing / broken summary only costs the traffic figure
        pass
    return None, None


def head_train_row(dev, hbm) -> dict:
    import torch.nn as nn
    from vpr_amd import ops
    D, hidden, n_out, Bt, N = D_DESC, 512, 2, 16, 1024
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1)
    Y = torch.randn(N, n_out, device=dev, generator=g)
    order = torch.randperm(N, device=dev, generator=g).to(torch.int32)
    torch.manual_seed(5)
    th = nn.Sequential(nn.Linear(D, hidden), nn.ReLU(), nn.Linear(hidden, n_out)).to(dev)
    W = [p.detach().clone() for p in (th[0].weight, th[0].bias, th[2].weight, th[2].bias)]
    m, v = ops.head_train_state(W[0], W[2])
    cnt = [1]

    def hip_epoch():
        ops.head_train_epoch(X, Y, order, Bt, *W, m, v, cnt[0])
        cnt[0] += N // Bt

    ms = _avg_ms(hip_epoch, n=10, warm=2) / (N // Bt)
    opt = torch.optim.AdamW(th.parameters(), lr=1e-5)
    ol = order.long()

    def torch_epoch():
        for lo in range(0, 16 * Bt, Bt):
            idx = ol[lo:lo + Bt]
            loss = nn.functional.mse_loss(th(X[idx]), Y[idx])
            opt.zero_grad()
            loss.backward()
            opt.step()

    ms_t = _avg_ms(torch_epoch, n=5, warm=1) / 16
    by = 7 * hidden * D * 4 + 2 * Bt * D * 4
    traffic, tsrc = None, None
    try:                                                        # PMC traffic per step, only if measured on this kernel source
        import hashlib
        with open(os.path.join(ROOT, "profiles", "r03_head_train_pmc.json")) as f:
            pj = json.load(f)
        with open(os.path.join(ROOT, pj["source"]), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest()[:16] == pj["source_sha16"]:
                traffic, tsrc = pj["hbm_bytes_per_step"], f"profiles/r03_head_train_pmc.json (source_sha16 {pj['source_sha16']})"
    except Exception:                                           # noqa: BLE001
        pass
    return dict(entry="vpr_head_train_epoch (per batch: forward, MSELoss, backward, AdamW — three launches)", traffic=traffic,
                traffic_source=tsrc,
                shape=f"B={Bt} D={D} hidden={hidden} n_out={n_out} f32", torch_autograd_adamw_ms=ms_t, speedup_vs_torch=ms_t / ms,
                **hbm(by, ms))


def kernel_rows(dev, ext, head, images, shard_bf16, a, salad_step_ms=None) -> dict:
    """Roofline rows of the other hand-written stages (BASELINE configs 2, 4, 5), each timed on its own after the timed
    region: algorithmic FLOPs / bytes over the average device time of the whole entry point."""
    from vpr_amd import _lib, ops
    from vpr_amd.retrieval import GraphedRetrieval, ShardedGallery
    rows = {}
    B, C = images.shape[0], ext.backbone.embed_dim
    hbm = lambda by, ms: {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": by / ms / 1e6 / HBM_PEAK_GBPS, "ms": ms, "algorithmic_bytes": by}
    # SALAD (config 2): both token MLPs, Sinkhorn, aggregation, norms — MFMA-bound dense stage + latency-bound Sinkhorn.
    # Two rows: `salad_aggregate` = the stage as every step of the timed region ran it (HIP events on the launch stream
    # around the aggregation: score + cluster MLPs fused into one kernel + Sinkhorn kernel; the token MLP — 64 cls rows,
    # two 4 us weight streams — runs on the backbone's cls-row stream ~0.3 ms earlier, so it is off this critical path but
    # its FLOPs are counted); `salad_aggregate_onecall` = the one-call C entry point with all three stages in one stream.
    tokens = ext.backbone(images, split=True)
    ms1 = _avg_ms(lambda: ext.aggregator(tokens, want_bf16=True))
    fl = B * (2 * 256 * C * 1024 + 2 * 256 * 512 * 192 + 2 * C * 512 + 2 * 512 * 256 + 2 * 128 * 64 * 256)
    mfma = lambda ms: {"bound": "mfma", "achieved": fl / ms / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, "ms": ms, "algorithmic_flops": fl}
    rows["salad_aggregate_onecall"] = dict(entry="vpr_salad_aggregate_split (token MLP, fused score/cluster MLPs, Sinkhorn: one stream)",
                                           shape=f"B={B} n=256 C={C} m=64 l=128 t=256", **mfma(ms1))
    if salad_step_ms:
        rows["salad_aggregate"] = dict(entry="in-step: vpr_salad_stage_mlps + vpr_salad_stage_aggregate on the launch stream, "
                                             "vpr_salad_stage_token on the backbone's cls-row stream",
                                       shape=f"B={B} n=256 C={C} m=64 l=128 t=256", samples=len(salad_step_ms),
                                       **mfma(sum(salad_step_ms) / len(salad_step_ms)))
    else:
        rows["salad_aggregate"] = dict(rows["salad_aggregate_onecall"])
    # fused (lat, lon, sin, cos) head on the descriptor (configs 2/3): weight stream
    desc = ext.aggregator(tokens)
    W1, b1h, W2, b2h = head.pack()
    # the C-ABI wrapper directly: through torch.ops.vpr.pose_head the dispatcher's ~25 us of host time per call exceed the
    # kernel's 20 us, and a back-to-back timing loop would measure the host
    ms = _avg_ms(lambda: ops.pose_head(desc, W1, b1h, W2, b2h, 2))
    by = (W1.numel() + W2.numel()) * 4 + desc.numel() * 4
    rows["pose_head"] = dict(entry="vpr_pose_head_split", shape=f"B={B} D={D_DESC} hidden={W1.shape[0]} n_out=4", **hbm(by, ms))
    # Swin-B fused LN + mean-pool + 4-wide head (config 4): B=256, T=49 (224 px) and T=144 (384 px), H=1024, bf16
    g = torch.Generator(device=dev).manual_seed(4)
    for T in (49, 144):
        xs = torch.randn(256, T, 1024, device=dev, generator=g).to(torch.bfloat16)
        gm, bt = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
        Wh, bh = torch.randn(4, 1024, device=dev, generator=g) * 0.03, torch.zeros(4, device=dev)
        ms = _avg_ms(lambda: ops.ln_meanpool_head(xs, gm, bt, 1e-5, Wh, bh, 2, want_pooled=False))
        rows[f"ln_meanpool_head_T{T}"] = dict(entry="vpr_ln_meanpool_head", shape=f"B=256 T={T} H=1024 bf16, n_out=4 (sin,cos unit)",
                                              **hbm(xs.numel() * 2, ms))
        del xs
    # head-only fine-tuning step on cached descriptors: the reference's head (8448 -> 512 -> 2) at its batch
    # size 16 (dinov2salad_finetuning.py:29-31,89,95-96); one epoch-call of 64 batches per sample; PyTorch autograd + AdamW on
    # the same GPU timed beside it (what the reference's loop runs per batch once the frozen backbone is taken out)
    rows["head_train_step"] = head_train_row(dev, hbm)
    # whole bf16 kNN call on the bench gallery (config 3 on one GPU) + the same retrieval replayed from a HIP graph
    if shard_bf16 is not None:
        N = shard_bf16.shape[0]
        q = torch.nn.functional.normalize(torch.randn(B, D_DESC, device=dev, generator=g), dim=1).to(torch.bfloat16)
        ws = ops.knn_workspace(B, N, D_DESC, a.k, dev)
        ms = _avg_ms(lambda: ops.knn_topk(q, shard_bf16, a.k, 0, ws))
        by = N * D_DESC * 2 + B * D_DESC * 2 + B * a.k * 8
        rows["knn_topk_bf16"] = dict(entry="vpr_knn_topk_checked", shape=f"B={B} N={N} D={D_DESC} k={a.k}", **hbm(by, ms))
        gr = GraphedRetrieval(ShardedGallery(shard_bf16, N), B, a.k)
        ms_g = _avg_ms(lambda: gr(q))
        rows["knn_topk_bf16_graph_replay"] = dict(entry="GraphedRetrieval (hipGraph: scores + select + final)",
                                                  shape=f"B={B} N={N}", **hbm(by, ms_g))
        del gr, ws
        # what ONE rank of an 8-GPU job runs per step on this gallery: the 512 all-gathered queries x its N/8-row shard
        # (MFMA GEMM route: gemm256_kernel, 2 K slices)
        if N >= 8 * 1024:
            Bg, Ns = 8 * B, N // 8
            qg = torch.nn.functional.normalize(torch.randn(Bg, D_DESC, device=dev, generator=g), dim=1).to(torch.bfloat16)
            sh = shard_bf16[:Ns]
            ws = ops.knn_workspace(Bg, Ns, D_DESC, a.k, dev)
            ev = []
            ms = _avg_ms(lambda: ops.knn_topk(qg, sh, a.k, 0, ws, score_events=ev))
            torch.cuda.synchronize()
            ms_k = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
            fl = 2.0 * Bg * Ns * D_DESC
            rows["knn_topk_bf16_gathered_8gpu_shard"] = {
                "entry": "vpr_knn_topk_scores_stage + vpr_knn_topk_select_stage", "shape": f"B={Bg} N={Ns} D={D_DESC} k={a.k}",
                "kernel": _lib.lib().vpr_knn_scores_kernel_name(0, Bg, Ns).decode(), "bound": "mfma",
                "achieved": fl / ms_k / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / ms_k / 1e9 / MFMA_BF16_PEAK_TFLOPS, "ms": ms_k, "whole_call_ms": ms, "algorithmic_flops": fl}
            del ws, qg
    # e4m3 gallery, 1M rows on one GPU (config 5's arithmetic and bytes; 8 GPUs would hold 125k rows each)
    N8 = a.fp8_rows
    if N8 > 0:
        g8 = torch.empty((N8, D_DESC), dtype=torch.uint8, device=dev)
        gs = torch.empty((N8,), dtype=torch.float32, device=dev)
        for lo in range(0, N8, 65536):
            n = min(65536, N8 - lo)
            x = torch.nn.functional.normalize(torch.randn(n, D_DESC, device=dev, generator=g), dim=1)
            g8[lo:lo + n], gs[lo:lo + n] = ops.quantize_fp8_rows(x)
            del x
        q8, qs = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(B, D_DESC, device=dev, generator=g), dim=1))
        ws = ops.knn_workspace(B, N8, D_DESC, a.k, dev)
        unc = torch.zeros(1, dtype=torch.int32, device=dev)
        ms = _avg_ms(lambda: ops.knn_topk_fp8(q8, qs, g8, gs, a.k, 0, ws, uncertified=unc), n=5)
        by = N8 * D_DESC + N8 * 4 + B * D_DESC + B * a.k * 8
        ev = []
        _avg_ms(lambda: ops.knn_topk_fp8(q8, qs, g8, gs, a.k, 0, ws, score_events=ev), n=5)       # the same call as its two stages
        ms_k = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
        rows["knn_topk_fp8"] = dict(entry="vpr_knn_topk_fp8_checked", shape=f"B={B} N={N8} D={D_DESC} e4m3 + per-row scale, k={a.k}",
                                    uncertified_queries=int(unc), score_kernel=_lib.lib().vpr_knn_scores_kernel_name(1, B, N8).decode(),
                                    score_kernel_ms=ms_k, score_kernel_frac=by / ms_k / 1e6 / HBM_PEAK_GBPS, **hbm(by, ms))
        if B == 64 and N8 == 1_000_000:
            t8, src8 = pmc_traffic("r03_knn8_pmc.json", rows["knn_topk_fp8"]["score_kernel"])
            rows["knn_topk_fp8"].update(score_kernel_traffic=t8, traffic_source=src8)
        gr = GraphedRetrieval(ShardedGallery(g8, N8, scales=gs), B, a.k)
        qb = torch.nn.functional.normalize(torch.randn(B, D_DESC, device=dev, generator=g), dim=1).to(torch.bfloat16)
        ms_g = _avg_ms(lambda: gr(qb), n=5)
        rows["retrieval_fp8_graph_replay"] = dict(entry="GraphedRetrieval (hipGraph: quantise queries + scores + select + final)",
                                                  shape=f"B={B} N={N8} e4m3", **hbm(by, ms_g))
        del gr, ws
        # config 5 as ONE rank of the 8-GPU job runs it: 512 gathered queries x a 125k-row e4m3 shard (gemm256_kernel<true>)
        if N8 >= 8 * 1024:
            Bg, Ns = 8 * B, N8 // 8
            q8g, qsg = ops.quantize_fp8_rows(torch.nn.functional.normalize(torch.randn(Bg, D_DESC, device=dev, generator=g), dim=1))
            ws = ops.knn_workspace(Bg, Ns, D_DESC, a.k, dev)
            ev = []
            ms = _avg_ms(lambda: ops.knn_topk_fp8(q8g, qsg, g8[:Ns], gs[:Ns], a.k, 0, ws, score_events=ev), n=5)
            torch.cuda.synchronize()
            ms_k = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
            fl = 2.0 * Bg * Ns * D_DESC
            rows["knn_topk_fp8_gathered_8gpu_shard"] = {
                "entry": "vpr_knn_topk_scores_stage + vpr_knn_topk_select_stage (e4m3)", "shape": f"B={Bg} N={Ns} D={D_DESC} k={a.k}",
                "kernel": _lib.lib().vpr_knn_scores_kernel_name(1, Bg, Ns).decode(), "bound": "mfma",
                "achieved": fl / ms_k / 1e9, "peak": MFMA_FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / ms_k / 1e9 / MFMA_FP8_PEAK_TFLOPS, "ms": ms_k, "whole_call_ms": ms, "algorithmic_flops": fl}
            del ws
        del g8, gs
    return rows


def config1_swin_tiny(dev) -> dict:
    """BASELINE config 1 (the reference's own CPU-runnable case, swin_transformer/swin_validation.py plumbing):
    Swin-Tiny 224x224 -> (lat, lon), batch 8.  CPU: HF SwinModel(SwinConfig()) (random init, seed 0) + Linear(768, 2) in
    f32 on the host cores, median of 10 calls after 3 warm-ups.  GPU: the same weights through
    vpr_amd.modules.SwinRegressionModel (PyTorch-ROCm backbone + vpr_ln_meanpool_head)."""
    import statistics
    from transformers import SwinConfig, SwinModel
    from vpr_amd.modules import SwinRegressionModel
    threads = min(os.cpu_count(), 16)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    backbone = SwinModel(SwinConfig()).eval()
    model = SwinRegressionModel(backbone).eval()
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ts = []
    with torch.no_grad():
        for i in range(13):
            t0 = time.perf_counter()
            pooled = backbone(pixel_values=x).pooler_output
            model.regressor(pooled)
            if i >= 3:
                ts.append(time.perf_counter() - t0)
    cpu = 8 / statistics.median(ts)
    gm = model.to(dev)
    xd = x.to(dev)
    with torch.no_grad():
        ms_eager = _avg_ms(lambda: gm(xd), n=10, warm=3)
    # batch 8 of Swin-T is launch-bound (~300 small kernels): the evaluation loops replay the forward from one HIP graph
    from vpr_amd.graphed import GraphedForward
    fwd = GraphedForward(gm)
    ms = _avg_ms(lambda: fwd(xd), n=10, warm=3)
    return {"workload": "Swin-Tiny 224 (random init) + Linear(768,2), batch 8, f32", "cpu_images_per_s": cpu, "cpu_threads": threads,
            "gpu_images_per_s": 8 / (ms * 1e-3), "gpu_ms_per_batch": ms, "gpu_ms_per_batch_eager": ms_eager,
            "gpu_path": "HIP-graph replay of the forward" if fwd.fallback_reason is None else f"eager ({fwd.fallback_reason})"}


def spawn_ranks(n: int) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>` as a child process group; returns its exit code.  The port is one the kernel just handed out."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    print(f"[bench] no launcher in the environment: spawning {n} rank(s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def collective_probe(gallery, batch: int, k: int, dev, n: int = 20) -> dict:
    """The exchange steps of one pipeline step on their own (every rank calls this): all-gather of the bf16 query
    descriptors, packed all-gather of the per-shard top-k, merge — HIP events on the launch stream, average of n.
    Also counts the ranks the process group really has (sum of ones over the group)."""
    from vpr_amd.retrieval import all_gather_topk
    ones = torch.ones(1, device=dev, dtype=torch.int32)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=gallery.group)
    q = torch.zeros((batch, D_DESC), dtype=torch.bfloat16, device=dev)
    v = torch.zeros((batch * gallery.world, k), dtype=torch.float32, device=dev)
    i = torch.zeros((batch * gallery.world, k), dtype=torch.int32, device=dev)

    def once():
        gallery.gather_queries(q)
        vs, is_ = all_gather_topk(v, i, gallery.world, gallery.group)
        gallery.engine.merge(vs, is_)
    return {"ranks_seen": int(ones.item()), "collective_ms_per_step": _avg_ms(once, n=n, warm=3),
            "collectives": "all_gather_into_tensor(queries bf16) + all_gather_into_tensor(top-k packed int32) + vpr_topk_merge"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--gallery", type=int, default=100_000, help="total gallery rows (sharded over ranks)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--arch", default="vit_large")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fold", action="store_true", help="keep LayerScale as separate multiplies (A/B)")
    ap.add_argument("--no-tune", action="store_true", help="no hipBLASLt kernel autotune (TunableOp) in warm-up")
    ap.add_argument("--no-resid-gemm", action="store_true", help="residual add fused into the LayerNorm instead of the proj/fc2 GEMM (A/B)")
    ap.add_argument("--no-split", action="store_true", help="backbone in the cls-first [B,257,C] row layout (M = 64.25 tile rows) instead of patch rows | cls rows (A/B)")
    ap.add_argument("--fuse-ln-cls", action="store_true", help="cls-row qkv/fc1 ride in the LayerNorm launch instead of separate 64-row launches (A/B)")
    ap.add_argument("--cls-before-gemm", action="store_true", help="cls-row launches before the library GEMM that shares their weights (A/B)")
    ap.add_argument("--knn-dtype", choices=["bf16", "fp8"], default="bf16",
                    help="gallery / query storage for the kNN stage: bf16 (headline) or e4m3 + per-row scale (BASELINE config 5 flavour)")
    ap.add_argument("--no-side-chain", action="store_true", help="cls-row kernels in the main stream instead of a side stream forked / joined once per block (A/B)")
    ap.add_argument("--side-sync", choices=("events", "light", "signals"), default=None,
                    help="fork / join of the cls side chain (A/B): torch events, HIP events without the system-scope fence, stream memory operations")
    ap.add_argument("--in-flight", type=int, default=1, help="independent batches in flight (one stream each); 1 = strictly sequential steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets 2 ranks rehearse on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank: still create the process group (RCCL) and run the two all-gathers + merge of the "
                         "sharded path (launch under torch.distributed.run --nproc-per-node 1, or bare: 127.0.0.1 rendezvous)")
    ap.add_argument("--graph-retrieval", action="store_true",
                    help="retrieval leg of every step = one HIP-graph replay of {query all-gather, shard search, top-k all-gather, "
                         "merge} (BASELINE config 5: --knn-dtype fp8 --gallery 1000000 --graph-retrieval on 8 GPUs)")
    ap.add_argument("--no-kernel-rows", action="store_true", help="skip the per-kernel roofline rows measured after the timed region")
    ap.add_argument("--fp8-rows", type=int, default=1_000_000, help="gallery rows of the e4m3 kNN row (BASELINE config 5 on one GPU)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.force_dist):
        # Bare `python bench.py --gpus N` (no launcher): become the launcher.  Nothing in this process has touched the GPU
        # yet (no HIP call, no torch.cuda.is_available()), so starting N fresh rank processes is safe; their stdout is
        # ours, i.e. rank 0's JSON line is the one line this command prints.
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    local_dev = local_rank % max(1, torch.cuda.device_count())     # == local_rank on a real N-GPU node
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:                            # bare `python bench.py --force-dist`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        with _StdoutToStderr():
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)     # RCCL over xGMI
            else:
                dist.init_process_group(a.backend, rank=rank, world_size=world)
            dist.barrier()                     # communicator creation (and RCCL's banner) happens here at the latest
            torch.cuda.synchronize()

    from vpr_amd import _lib, ops
    _lib.lib()
    from vpr_amd.modules import DinoV2Salad, FusedGeoPoseHead
    from vpr_amd.pipeline import VPRGeoPosePipeline
    from vpr_amd.retrieval import ShardedGallery, shard_bounds

    torch.manual_seed(0)                                   # identical weights on every rank
    ext32 = DinoV2Salad(a.arch).eval()
    for p in ext32.aggregator.parameters():                # SALAD weights ~ N(0, 0.02), dustbin 1
        if p.dim() > 0:
            nn.init.normal_(p, std=0.02)
    ext_state = {k: v.clone() for k, v in ext32.state_dict().items()}
    ext = ext32.to(dev).to(torch.bfloat16)
    ext.backbone.residual_in_gemm = not a.no_resid_gemm
    ext.backbone.hip_split = not a.no_split
    ext.backbone.fuse_ln_cls = a.fuse_ln_cls
    ext.backbone.cls_after_gemm = not a.cls_before_gemm
    ext.backbone.cls_side_chain = not a.no_side_chain
    if a.side_sync is not None:
        ext.backbone.side_sync = a.side_sync
    ext.backbone.auto_fold = not a.no_fold
    if not a.no_fold:
        ext.backbone.fold_layerscale()          # inference-only: two fewer elementwise passes per block
    pos = nn.Sequential(nn.Linear(D_DESC, 512), nn.ReLU(), nn.Linear(512, 2))
    ang = nn.Sequential(nn.Linear(D_DESC, 512), nn.ReLU(), nn.Linear(512, 2))
    head = FusedGeoPoseHead(pos.to(dev), ang.to(dev), normalize=True)
    head_cpu = [t.cpu() for t in head.pack()]

    lo, hi = shard_bounds(a.gallery, rank, world)
    shard = make_gallery_shard(hi - lo, 1 + rank, dev)
    if a.knn_dtype == "fp8":
        g8 = torch.empty((hi - lo, D_DESC), dtype=torch.uint8, device=dev)
        gs = torch.empty((hi - lo,), dtype=torch.float32, device=dev)
        for r0 in range(0, hi - lo, 65536):                                     # slabs: the f32 copy of a big shard is 4x its bf16 size
            q8, qs = ops.quantize_fp8_rows(shard[r0:r0 + 65536].float())
            g8[r0:r0 + 65536], gs[r0:r0 + 65536] = q8, qs
        gallery = ShardedGallery(g8, a.gallery, rank, world, scales=gs, force_collectives=a.force_dist)
    else:
        gallery = ShardedGallery(shard, a.gallery, rank, world, force_collectives=a.force_dist)
    pipe = VPRGeoPosePipeline(ext, head, gallery, a.k, graph_retrieval=a.graph_retrieval)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    images = torch.randn(a.batch, 3, 224, 224, device=dev, generator=g).to(torch.bfloat16)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    from vpr_amd.backbone import gemm_autotune
    if not a.no_tune:
        gemm_autotune(True, tuning=True, max_ms_per_gemm=int(os.environ.get("VPR_TUNE_MS", "30")))   # picks the backbone GEMM kernels during warm-up
    for _ in range(max(a.warmup, 1)):
        pipe.step(images)
    if not a.no_tune:
        torch.cuda.synchronize()
        gemm_autotune(True, tuning=False)       # timed region: replay only
    pipe.knn_events = []
    pipe.salad_events = []
    lanes = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(a.in_flight - 1)]
    if a.in_flight > 1:                       # every lane warms its own workspaces / side stream outside the timed region
        for s in lanes[1:]:
            s.wait_stream(lanes[0])
            with torch.cuda.stream(s):
                pipe.step(images)
        pipe.knn_events = []
        pipe.salad_events = []
    sync()
    t0 = time.perf_counter()
    for it in range(a.steps):
        if a.in_flight == 1:
            out = pipe.step(images)
        else:                                 # independent batches in flight on separate streams
            with torch.cuda.stream(lanes[it % a.in_flight]):
                out = pipe.step(images)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    knn_ms = sorted(e0.elapsed_time(e1) for e0, e1 in pipe.knn_events)
    knn_avg_s = sum(knn_ms) / len(knn_ms) * 1e-3
    pipe.knn_events = None
    salad_step_ms = [e0.elapsed_time(e1) for e0, e1 in pipe.salad_events]
    pipe.salad_events = None
    n_shard, bq = hi - lo, a.batch * world
    esz = 1 if a.knn_dtype == "fp8" else 2
    alg_bytes = n_shard * D_DESC * esz + bq * D_DESC * esz + bq * a.k * 8   # SURVEY §8d per query batch (s = 2 bf16, 1 fp8)
    # HBM traffic of the score kernel from the PMC counters (profiles/, collected as MI355X_MICROARCH.md §HBM prescribes):
    # quoted only if the summary was measured on THIS kernel source and names the kernel this run launched.
    traffic, traffic_source = None, None
    score_kernel = _lib.lib().vpr_knn_scores_kernel_name(int(a.knn_dtype == "fp8"), bq, n_shard).decode()
    if world == 1 and a.batch == 64 and not a.graph_retrieval:
        if a.gallery == 100_000 and a.knn_dtype == "bf16":
            traffic, traffic_source = pmc_traffic("r03_knn_pmc.json", score_kernel)
        elif a.gallery == 1_000_000 and a.knn_dtype == "fp8":
            traffic, traffic_source = pmc_traffic("r03_knn8_pmc.json", score_kernel)

    # knn_avg_s = the score stage alone (HIP events between the two stages of vpr_knn_topk*, same kernels as the one-call form)
    if a.graph_retrieval:
        # a graph has no seam for events: the whole replay (collectives, quantisation, scores, select, merge) is the timed unit
        roofline = {"bound": "hbm", "kernel": f"hipGraph replay of the retrieval leg ({'2 all-gathers + merge + ' if gallery.collective else ''}"
                                              f"{'query quantisation + ' if a.knn_dtype == 'fp8' else ''}{score_kernel} + select)",
                    "achieved": alg_bytes / knn_avg_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": alg_bytes / knn_avg_s / 1e9 / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                    "kernel_ms": knn_avg_s * 1e3, "algorithmic_bytes": alg_bytes}
    elif bq <= 64:
        roofline = {"bound": "hbm", "kernel": score_kernel,
                    "achieved": alg_bytes / knn_avg_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": alg_bytes / knn_avg_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                    "kernel_ms": knn_avg_s * 1e3, "algorithmic_bytes": alg_bytes}
    else:
        # more than one 64-query tile per shard scan (multi-GPU: all-gathered queries): the score tile runs as an MFMA
        # GEMM (gemm256_kernel / gemm_nt_kernel, K-split for small shards) — 2*bq FLOP per gallery row element
        flops = 2.0 * bq * n_shard * D_DESC
        peak = MFMA_FP8_PEAK_TFLOPS if a.knn_dtype == "fp8" else MFMA_BF16_PEAK_TFLOPS
        roofline = {"bound": "mfma", "kernel": f"{score_kernel} (score tile, query batch > 64)",
                    "achieved": flops / knn_avg_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                    "frac": flops / knn_avg_s / 1e12 / peak, "traffic": None,
                    "kernel_ms": knn_avg_s * 1e3, "algorithmic_flops": flops}

    # Planted-positive Recall@1 THROUGH THE SHARDED PATH (outside the timed region; every rank takes part): rank r plants
    # its B queries next to rows of ITS OWN shard, so the all-gathered batch has its positives spread over all shards;
    # every query is searched on every shard, the per-shard top-k are all-gathered and merged, and each rank checks its own
    # rows of the merged answer against the global index it planted.  The reported value is the mean over all ranks.
    gp = torch.Generator(device=dev).manual_seed(7 + rank)
    pos_idx = torch.randint(0, n_shard, (a.batch,), device=dev, generator=gp)
    qn = torch.nn.functional.normalize(shard[pos_idx].float() + 0.1 * torch.randn(a.batch, D_DESC, device=dev, generator=gp), dim=1)
    _, ti = gallery.search_local_queries(qn.to(torch.bfloat16), 1)
    hits = (ti[:, 0].long() == pos_idx + lo).double().sum().reshape(1)
    if use_dist:
        dist.all_reduce(hits, op=dist.ReduceOp.SUM)
    recall1 = float(hits.item()) / (a.batch * world)
    uncertified = gallery.uncertified_queries()                 # summed over the group when the gallery is sharded (a collective)
    dist_info = {"process_group": (a.backend if use_dist else None), "collectives_in_step": gallery.collective,
                 "ranks_seen": world if not use_dist else None, "collective_ms_per_step": None}
    if gallery.collective:
        dist_info.update(collective_probe(gallery, a.batch, a.k, dev))

    if rank == 0:

        # per-stage device time (one extra step each, outside the timed region)
        def stage_ms(fn, n=3):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                r = fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, r
        stages = {}
        stages["backbone_ms"], tokens = stage_ms(lambda: ext.backbone(images, split=True))
        stages["salad_ms"], (desc, desc16) = stage_ms(lambda: ext.aggregator(tokens, want_bf16=True))
        if world == 1:
            stages["knn_ms"], _ = stage_ms(lambda: gallery.search(desc16, a.k))
        stages["head_ms"], _ = stage_ms(lambda: head(desc))

        res = {
            "metric": "images/sec end-to-end (backbone->SALAD->kNN->pose)",
            "value": a.steps * a.batch * world / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.knn_dtype == "bf16" else "bf16 (backbone, SALAD) + fp8 e4m3 kNN", "data": "synthetic",
            "config": {"workload": f"DINOv2 {a.arch}/14 (random init) + SALAD + {a.knn_dtype} kNN k={a.k} over a "
                                   f"{a.gallery}-row x {D_DESC} synthetic gallery sharded {world} way(s) + fused "
                                   f"(lat,lon,sin,cos) head; 3x224x224 bf16 images",
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world, "gallery_rows": a.gallery,
                       "k": a.k, "parallelism": f"dp{world}+gallery-shard{world}", "batches_in_flight": a.in_flight,
                       "graph_retrieval": bool(a.graph_retrieval)},
            "roofline": roofline,
            "recall_at_1": recall1,
            "recall_path": "search_local_queries: query all-gather -> shard search on every rank -> top-k all-gather -> merge; "
                           "positives planted in every shard" if gallery.collective else "local search (one shard, no collectives)",
            "uncertified_queries": uncertified,    # kNN answers the device could not certify exact, all shards (0 expected)
            "stages": stages,
            "dist": dist_info,
        }
        if world == 1 and not a.no_kernel_rows:
            try:                                                # auxiliary rows never cost the run its headline line
                res["kernels"] = kernel_rows(dev, ext, head, images, shard if a.knn_dtype == "bf16" else None, a, salad_step_ms)
            except Exception as e:                              # noqa: BLE001
                res["kernels"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            sample = shard.cpu()                                # whole gallery: the sample is one full step
            res["cpu_baseline"] = cpu_baseline(ext_state, a.arch, head_cpu, images.cpu(), sample, a.gallery, a.k)
            try:
                res["cpu_baseline"]["config1_swin_tiny"] = config1_swin_tiny(dev)
            except Exception as e:                              # noqa: BLE001  (e.g. transformers missing on the box)
                res["cpu_baseline"]["config1_swin_tiny"] = {"error": f"{type(e).__name__}: {e}"}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
