"""GPU parity of the regression / angle heads against oracle/heads.py (fp64) — outputs within
1e-4 (north star tolerance on the standardised head output; observed ~1e-6)."""
import pytest
import torch

from oracle import heads as oheads

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _linear_init(out_f, in_f, g):
    bound = 1.0 / in_f ** 0.5
    return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound, (torch.rand(out_f, generator=g) * 2 - 1) * bound)


@pytest.mark.parametrize("B,D,hidden,n_out,off", [
    (64, 8448, 512, 2, -1),     # DINOv2RegressionModel.regressor
    (64, 8448, 512, 4, 2),      # fused (lat, lon, sin, cos)
    (64, 8448, 1024, 4, 2),     # bench.py's head: FusedGeoPoseHead of two 512-wide MLPs (hidden 2 x 512)
    (256, 1024, 512, 2, -1),    # Swin-Base MLP head
    (7, 768, 384, 2, 0),        # sin/cos MLP head, ragged batch
    (1, 64, 32, 1, -1),
])
@pytest.mark.parametrize("split", ["counters", "frag", True, False])
def test_mlp_head(dev, tune, B, D, hidden, n_out, off, split):
    """"counters": the single-launch kernel (fragment-order (hi, lo) planes, split-K finished by arrival counters);
    "frag": the same kernel writing slabs only + an epilogue launch; True: the two-launch split form on row-major planes
    (default); False: exact-f32 MFMA.  All must sit far inside the 1e-4 tolerance: the split paths' error budget is
    2^-16 per product, ~1e-7 on the output."""
    if split == "counters":
        tune("VPR_POSE_VARIANT", 1)
    from vpr_amd import ops
    g = torch.Generator().manual_seed(B + D)
    x = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1) if D == 8448 else torch.randn(B, D, generator=g)
    W1, b1 = _linear_init(hidden, D, g)
    W2, b2 = _linear_init(n_out, hidden, g)
    ref = oheads.mlp_head(x, W1, b1, W2, b2, off)
    out = ops.pose_head(x.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), off, split=bool(split),
                        fused=split in ("counters", "frag")).cpu().double()
    err = (out - ref).abs().max().item()
    print("mlp head err", err)
    assert err < TOL
    assert err < 2e-5 * max(1.0, ref.abs().max().item())          # f32-level, both paths (O(1) randn inputs: ~1e-5)


@pytest.mark.parametrize("B,D,n_out,off", [(8, 768, 2, -1), (8, 768, 2, 0), (33, 1024, 4, 2)])
def test_linear_head(dev, B, D, n_out, off):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(D + n_out)
    x = torch.randn(B, D, generator=g)
    W2, b2 = _linear_init(n_out, D, g)
    ref = oheads.mlp_head(x, None, None, W2, b2, off)
    out = ops.pose_head(x.to(dev), None, None, W2.to(dev), b2.to(dev), off).cpu().double()
    assert (out - ref).abs().max().item() < TOL


def test_head_is_deterministic(dev):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 8448, generator=g).to(dev)
    W1, b1 = _linear_init(512, 8448, g)
    W2, b2 = _linear_init(4, 512, g)
    args = [t.to(dev) for t in (W1, b1, W2, b2)]
    a = ops.pose_head(x, *args, 2)
    b = ops.pose_head(x, *args, 2)
    assert torch.equal(a, b)


def test_fused_head_counters_survive_many_calls_and_shapes(dev, tune):
    """vpr_pose_head_fused: the arrival counters at the head of the workspace are left zero by every call, so calls of
    different shapes can share one workspace back to back (the wrapper zero-fills a workspace only when it allocates it);
    every call is bitwise reproducible (fixed summation orders at both counter levels) and within TOL of the f64 oracle.
    Ragged batches (rows of the last 64-row tile masked), hidden not a multiple of 64 (masked columns), D = 768."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(11)
    cases = []
    for B, D, hidden, n_out, off in ((64, 8448, 1024, 4, 2), (7, 768, 384, 2, 0), (130, 1024, 512, 2, -1), (1, 64, 48, 1, -1),
                                     (256, 1024, 512, 8, -1)):
        x = torch.randn(B, D, generator=g)
        W1, b1 = _linear_init(hidden, D, g)
        W2, b2 = _linear_init(n_out, hidden, g)
        ref = oheads.mlp_head(x, W1, b1, W2, b2, off)
        cases.append(([t.to(dev) for t in (x, W1, b1, W2, b2)], off, ref))
    tune("VPR_POSE_VARIANT", 1)                                     # the arrival-counter form
    first = {}
    for rep in range(6):
        for i, (args, off, ref) in enumerate(cases):
            out = ops.pose_head(*args, off, fused=True)
            if rep == 0:
                first[i] = out.clone()
                assert (out.cpu().double() - ref).abs().max().item() < TOL, i
            else:
                assert torch.equal(out, first[i]), (rep, i)
    torch.cuda.synchronize()
    ws = ops.workspace("pose_fused", 256, dev)                      # the shared workspace: its 4 KB of counter words are all zero again
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0
    # a batch so large that its counters would not fit the fixed 4 KB goes through the two-launch form by itself
    B, D, hidden = 64 * 70, 64, 1024
    x, (W1, b1), (W2, b2) = torch.randn(B, D, generator=g), _linear_init(hidden, D, g), _linear_init(2, hidden, g)
    out = ops.pose_head(x.to(dev), W1.to(dev), b1.to(dev), W2.to(dev), b2.to(dev), fused=True)
    assert (out.cpu().double() - oheads.mlp_head(x, W1, b1, W2, b2)).abs().max().item() < TOL


def test_zero_pair_normalise_eps(dev):
    """F.normalize eps path: a zero (sin, cos) pair stays zero instead of NaN."""
    from vpr_amd import ops
    x = torch.zeros(2, 64)
    W2, b2 = torch.zeros(2, 64), torch.zeros(2)
    out = ops.pose_head(x.to(dev), None, None, W2.to(dev), b2.to(dev), 0).cpu()
    assert torch.equal(out, torch.zeros(2, 2))


@pytest.mark.parametrize("B,T,H,dtype", [
    (8, 49, 768, torch.float32),      # Swin-T
    (4, 144, 1024, torch.bfloat16),   # Swin-B 384
    (256, 49, 1024, torch.bfloat16),  # BASELINE config 4
    (256, 144, 1024, torch.bfloat16), # config 4 on the 384-px Swin-B the reference actually used (swin_attempt_2.py:32-33)
    (3, 5, 512, torch.float32),
    (2, 1, 1536, torch.bfloat16),
])
def test_ln_meanpool_head(dev, B, T, H, dtype):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(T + H)
    x = (torch.randn(B, T, H, generator=g) * 1.5 + 0.3).to(dtype)
    gamma = 1 + 0.1 * torch.randn(H, generator=g)
    beta = 0.1 * torch.randn(H, generator=g)
    Wh, bh = _linear_init(4, H, g)
    pooled_ref, out_ref = oheads.ln_meanpool_head(x, gamma, beta, 1e-5, Wh, bh, 2)
    pooled, out = ops.ln_meanpool_head(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, Wh.to(dev), bh.to(dev), 2)
    assert (pooled.cpu().double() - pooled_ref).abs().max().item() < 2e-5
    assert (out.cpu().double() - out_ref).abs().max().item() < TOL
    pooled_only, none = ops.ln_meanpool_head(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5)
    assert none is None and torch.equal(pooled_only, pooled)


@pytest.mark.parametrize("M,C,pdtype", [(16448, 1024, torch.bfloat16), (514, 384, torch.bfloat16), (7, 768, torch.float32),
                                        (3, 2048, torch.float32)])
def test_layernorm_bf16_matches_torch(dev, M, C, pdtype):
    """Backbone LayerNorm kernel vs an f64 LayerNorm of the same bf16 inputs (output within one bf16 ulp)."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(torch.bfloat16)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(pdtype)
    beta = (0.1 * torch.randn(C, generator=g)).to(pdtype)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-6)
    y = ops.layernorm_bf16(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6).cpu()
    err = (y.double() - ref).abs()
    assert (err <= 2 ** -8 * ref.abs() + 1e-6).all()      # within one bf16 rounding of the exact value


def test_add_layernorm_equals_add_then_layernorm(dev):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 1024, generator=g).to(torch.bfloat16).to(dev)
    r = (torch.randn(1000, 1024, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    gamma = (1 + 0.1 * torch.randn(1024, generator=g)).to(torch.bfloat16).to(dev)
    beta = (0.1 * torch.randn(1024, generator=g)).to(torch.bfloat16).to(dev)
    s, y = ops.add_layernorm_bf16(x, r, gamma, beta, 1e-6)
    assert torch.equal(s, x + r)                                   # bf16 add, rounded once, as torch
    assert torch.equal(y, ops.layernorm_bf16(x + r, gamma, beta, 1e-6))


@pytest.mark.parametrize("B,Cin,H,W,P,kpad,lead", [(3, 3, 224, 224, 14, 640, 1), (2, 3, 224, 224, 14, 592, 0),
                                                    (1, 1, 32, 64, 8, 64, 2), (2, 3, 64, 32, 16, 768, 1),
                                                    (2, 3, 14, 56, 7, 152, 1)])      # odd patch: 2-byte LDS path
def test_patchify_matches_unfold(dev, B, Cin, H, W, P, kpad, lead):
    """Bit-exact vs F.unfold (the conv's im2col order c*P*P + i*P + j), zero cls rows and K padding."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, Cin, H, W, generator=g).to(torch.bfloat16)
    out = ops.patchify_bf16(img.to(dev), P, kpad, lead).cpu()
    n, K = (H // P) * (W // P), Cin * P * P
    ref = torch.nn.functional.unfold(img.float(), P, stride=P).transpose(1, 2).to(torch.bfloat16)   # [B, n, K]
    out = out.view(B, lead + n, kpad)
    assert torch.equal(out[:, lead:, :K], ref)
    assert not out[:, :lead].any() and not out[:, :, K:].any()


def test_bias_layernorm_matches_torch(dev):
    """LayerNorm(f32(x) + pre_bias): the offset is added in f32 before the statistics."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(515, 1024, generator=g).to(torch.bfloat16)
    pb = torch.randn(1024, generator=g) * 0.5
    gamma = (1 + 0.1 * torch.randn(1024, generator=g)).to(torch.bfloat16)
    beta = (0.1 * torch.randn(1024, generator=g)).to(torch.bfloat16)
    ref = torch.nn.functional.layer_norm(x.float() + pb, (1024,), gamma.float(), beta.float(), 1e-6)
    y = ops.bias_layernorm_bf16(x.to(dev), pb.to(dev), gamma.to(dev), beta.to(dev), 1e-6).cpu().float()
    assert (y - ref).abs().max().item() < 0.02          # one bf16 rounding of values up to ~4
    zero = ops.bias_layernorm_bf16(x.to(dev), torch.zeros(1024, device=dev), gamma.to(dev), beta.to(dev), 1e-6)
    assert torch.equal(zero, ops.layernorm_bf16(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6))


@pytest.mark.parametrize("gelu,C,N,n_cls,M", [(False, 1024, 3072, 64, 16448), (True, 1024, 4096, 64, 16448),
                                                (False, 384, 1152, 3, 771), (True, 768, 72, 70, 500)])
def test_bias_layernorm_cls_linear(dev, gelu, C, N, n_cls, M):
    """The fused launch == vpr_bias_layernorm_bf16 on all rows (bit-exact) + the skinny linear on the
    LayerNorm of the cls rows (same bf16 operands; tolerance = accumulation order + one bf16 rounding)."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(C + N)
    x = (torch.randn(M, C, generator=g) * 2).to(torch.bfloat16).to(dev)
    pb = (torch.randn(C, generator=g) * 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, C, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    lb = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    row0 = M - n_cls
    # the cls rows come out of a skinny accumulate that also leaves their statistics partials
    prev_in = (torch.randn(n_cls, 64, generator=g)).to(torch.bfloat16).to(dev)
    prev_w = (torch.randn(C, 64, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    rs = torch.empty((C // 16, n_cls, 2), dtype=torch.float32, device=dev)
    ops.skinny_linear_bf16(prev_in, prev_w, None, x[row0:], 2, pb, rs)
    v = x[row0:].float() + pb                                         # what the LayerNorm reads
    blocks = v.view(n_cls, C // 16, 16)
    assert torch.allclose(rs[:, :, 0].T, blocks.mean(-1), atol=1e-5)
    assert torch.allclose(rs[:, :, 1].T, ((blocks - blocks.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=1e-4, atol=1e-4)
    out = torch.full((n_cls, N), 7.0, dtype=torch.bfloat16, device=dev)
    consts = ops.ClsLinearConsts.build(w, lb, gamma, beta, pb)
    y = ops.bias_layernorm_cls_linear_bf16(x, pb, gamma, beta, 1e-6, row0, rs, consts, out, gelu=gelu)
    y_ref = ops.bias_layernorm_bf16(x, pb, gamma, beta, 1e-6)
    assert torch.equal(y, y_ref)
    # reference: the exact LayerNorm of the cls rows (f64) through the exact linear layer; the fused path rounds
    # W*gamma to bf16 instead of the normalised activations: same size of error as the unfused path
    ln = torch.nn.functional.layer_norm(v.double(), (C,), gamma.double(), beta.double(), 1e-6)
    acc = ln @ w.double().T + lb.double()
    ref = torch.nn.functional.gelu(acc, approximate="tanh") if gelu else acc
    err = (out.double() - ref).abs().max().item()
    unfused = y_ref[row0:].double() @ w.double().T + lb.double()
    unfused = torch.nn.functional.gelu(unfused, approximate="tanh") if gelu else unfused
    err_unfused = (unfused.to(torch.bfloat16).double() - ref).abs().max().item()
    assert err < max(2.0 * err_unfused, 8e-3 * max(1.0, ref.abs().max().item())), (err, err_unfused)


@pytest.mark.parametrize("mode", ["split_side", "split", "split_unfused", "resid_gemm", "add_ln"])
def test_backbone_hip_path_matches_block_loop(dev, mode):
    """The HIP backbone paths (split row layout with patchify embedding; cls-first layout with the
    residual add inside the proj/fc2 GEMMs + deferred biases; cls-first with fused add+LN) give the
    same tokens as the plain block loop."""
    from vpr_amd.backbone import DinoV2
    torch.manual_seed(0)
    m = DinoV2("vit_small").to(dev).to(torch.bfloat16).eval()
    m.hip_split = mode.startswith("split")
    m.cls_side_chain = mode == "split_side"
    m.fuse_ln_cls = mode == "split"
    m.residual_in_gemm = mode != "add_ln"
    for b in m.blocks:
        torch.nn.init.normal_(b.ls1, std=0.3)
        torch.nn.init.normal_(b.ls2, std=0.3)
    m.fold_layerscale()
    x = torch.randn(3, 3, 224, 224, device=dev, dtype=torch.bfloat16)
    import copy
    m32 = copy.deepcopy(m).float()                     # same (bf16-rounded, LayerScale-folded) weights, f32 math

    def block_loop(model, inp):
        t = model.patch_embed(inp).flatten(2).transpose(1, 2)
        t = torch.cat([model.cls_token.expand(3, -1, -1), t], dim=1) + model.pos_embed
        for blk in model.blocks:
            t = blk(t)
        return torch.nn.functional.layer_norm(t.float(), (384,), model.norm.weight.float(), model.norm.bias.float(), 1e-6)

    with torch.no_grad():
        fast = m(x)
        slow = block_loop(m, x)                        # bf16 activations, PyTorch ops only
        ref = block_loop(m32, x.float())               # f32 activations
    assert fast.shape == (3, 257, 384)
    # two bf16 pipelines differ from each other by up to the sum of their errors; judge each against f32:
    # the HIP path must not be worse than the plain bf16 block loop (RMS: the max over 3e5 values is noise)
    rms = lambda d: d.float().pow(2).mean().sqrt().item()
    err_fast, err_slow = rms(fast.float() - ref), rms(slow - ref)
    assert err_fast < 1.15 * err_slow and err_fast < 0.02 * rms(ref), (err_fast, err_slow, rms(ref))
    assert (fast.float() - ref).abs().max().item() < 2.0 * (slow - ref).abs().max().item()


def test_patch_embed_hip_matches_conv(dev):
    """patchify + GEMM + static offsets == conv + cls cat + position add (to bf16 rounding of O(1) values)."""
    from vpr_amd import ops
    from vpr_amd.backbone import DinoV2
    torch.manual_seed(1)
    m = DinoV2("vit_small").to(dev).to(torch.bfloat16).eval()
    torch.nn.init.normal_(m.cls_token, std=0.5)
    m.fold_layerscale()
    x = torch.randn(2, 3, 224, 224, device=dev, dtype=torch.bfloat16)
    assert m._hip_split_ok(x)
    with torch.no_grad():
        w, off = m._embed_consts(x)
        raw = ops.patchify_bf16(x, 14, w.shape[1], 0) @ w.t()                       # [2*256, 384]
        body = (raw.float() + off[:512].float()).view(2, 256, 384)
        cls = off[512:].float()
        m32 = m.float()
        t = m32.patch_embed(x.float()).flatten(2).transpose(1, 2)
        ref = torch.cat([m32.cls_token.expand(2, -1, -1), t], dim=1) + m32.pos_embed
    tol = 0.02 * max(1.0, ref.abs().max().item())
    assert (body - ref[:, 1:]).abs().max().item() < tol and (cls - ref[:, 0]).abs().max().item() < tol


def test_cls_side_chain_is_bit_identical_to_in_stream_path(dev):
    """The cls rows on their own stream (one fork after each attention, one join before the next): same kernels
    on the same data, so the tokens must equal the single-stream path bit for bit — run twice to catch a race."""
    from vpr_amd.backbone import DinoV2
    torch.manual_seed(3)
    m = DinoV2("vit_small").to(dev).to(torch.bfloat16).eval()
    m.fold_layerscale()
    x = torch.randn(5, 3, 224, 224, device=dev, dtype=torch.bfloat16)
    m.cls_side_chain = False
    ref = m(x, split=True)
    m.cls_side_chain = True
    for _ in range(3):
        got = m(x, split=True)
        assert torch.equal(got.patch, ref.patch) and torch.equal(got.cls, ref.cls)


def test_two_streams_drive_the_extractor_concurrently(dev):
    """Two batches in flight on two streams (each with its own workspaces, raw-token buffer and cls side
    stream) give the descriptors of the sequential run, bit for bit."""
    from vpr_amd.modules import DinoV2Salad
    torch.manual_seed(4)
    ext = DinoV2Salad("vit_small").to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    xs = [torch.randn(6, 3, 224, 224, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    ref = [ext(x).clone() for x in xs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        outs = [None, None]
        for j, (x, s) in enumerate(zip(xs, (s1, s2))):
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                outs[j] = ext(x)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])


def test_split_tokens_roundtrip_and_salad_split(dev):
    """forward(split=True).joined() == forward(); SALAD on the pair == SALAD on the joined tensor (bit-exact:
    same kernels, only the row addressing differs)."""
    from vpr_amd.modules import DinoV2Salad
    torch.manual_seed(2)
    ext = DinoV2Salad("vit_small").to(dev).to(torch.bfloat16).eval()
    ext.backbone.fold_layerscale()
    x = torch.randn(3, 3, 224, 224, device=dev, dtype=torch.bfloat16)
    st = ext.backbone(x, split=True)
    assert st.patch.shape == (3, 256, 384) and st.cls.shape == (3, 384)
    joined = ext.backbone(x)
    assert torch.equal(joined, st.joined()) and torch.equal(joined[:, 0], st.cls)
    d_split = ext.aggregator(st)
    d_join = ext.aggregator(joined)
    assert torch.equal(d_split, d_join)
    assert torch.equal(ext(x), d_split)


def test_attention_split_layout_matches_contiguous(dev):
    """vpr_attention_qkv_split_bf16 on [patch rows | cls rows] == vpr_attention_qkv_bf16 on the same tokens cls-last."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(3)
    B, T, H = 5, 257, 6
    qkv = (torch.randn(B, T, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16).to(dev)     # token T-1 plays the cls role
    ref = ops.attention_qkv_bf16(qkv, H)                                                       # [B, T, C]
    split_in = torch.cat([qkv[:, :T - 1].reshape(B * (T - 1), -1), qkv[:, T - 1]], dim=0).contiguous()
    out = ops.attention_qkv_split_bf16(split_in, B, T, T - 1, H)
    assert torch.equal(out[:B * (T - 1)].view(B, T - 1, -1), ref[:, :T - 1])
    assert torch.equal(out[B * (T - 1):], ref[:, T - 1])


@pytest.mark.parametrize("B,T,H", [(64, 257, 16), (2, 257, 6), (3, 100, 2), (1, 288, 1), (2, 17, 3)])
def test_attention_matches_f64_reference(dev, B, T, H):
    """Short-sequence attention kernel vs softmax(q k^T / 8) v in f64 on the same bf16 inputs
    (P is rounded to bf16 before P·V, as in every flash kernel: tolerance 2e-2 of the value range)."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B, T, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16)
    q, k, v = qkv.double().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v          # [B,H,T,64]
    ref = ref.transpose(1, 2).reshape(B, T, H * 64)
    out = ops.attention_qkv_bf16(qkv.to(dev), H).cpu().double()
    err = (out - ref).abs().max().item()
    print("attention max err", err)
    assert err < 2e-2
    sdpa = torch.nn.functional.scaled_dot_product_attention(
        *[t.to(dev) for t in qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)]).transpose(1, 2).reshape(B, T, H * 64)
    # and within two bf16 ulps (outputs reach |x| ~ 6, ulp 0.031) of PyTorch's own kernel
    assert (out - sdpa.cpu().double()).abs().max().item() < 7e-2


def test_pose_head_plane_cache_survives_a_recycled_address(dev):
    """The (hi, lo) bf16 planes of W1 are cached per weight storage.  A weight that is freed and another of the same
    shape allocated at the same address (the caching allocator does exactly that) must not hit the old planes:
    the cache entry pins its weight (round-2 fix; the failure showed up as a 1e-2 error in test_mlp_head)."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(77)
    B, D, hidden = 8, 1024, 64
    x = torch.randn(B, D, generator=g)
    W2, b2 = _linear_init(2, hidden, g)
    ptrs = set()
    for trial in range(4):
        W1, b1 = _linear_init(hidden, D, g)
        W1d = W1.to(dev)
        ptrs.add(W1d.data_ptr())
        out = ops.pose_head(x.to(dev), W1d, b1.to(dev), W2.to(dev), b2.to(dev), -1, split=True).cpu().double()
        ref = oheads.mlp_head(x, W1, b1, W2, b2, -1)
        assert (out - ref).abs().max().item() < TOL, trial
        del W1d
    # whether or not the allocator reused the address this time, every trial had to be right
    assert len(ptrs) >= 1
