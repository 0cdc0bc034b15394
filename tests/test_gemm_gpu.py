"""MFMA GEMM (vpr_gemm_nt_bf16) against an fp64 reference on identical bf16 operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,relu,out_bf16", [
    (128, 128, 64, False, False),
    (256, 64, 128, True, False),
    (300, 200, 192, True, True),      # ragged M, N
    (64, 512, 1024, True, True),      # token MLP layer 1 shape
    (1024, 1024, 1024, True, True),   # fused score+cluster layer 1 (4 images)
    (1024, 64, 512, False, False),    # score layer 2
    (1024, 128, 512, False, False),   # cluster layer 2
    (5, 3, 64, False, False),
])
def test_gemm_nt(dev, M, N, K, relu, out_bf16):
    from vpr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    # asymmetric, non-constant bias catches row/col swaps
    bias = torch.linspace(-1, 1, N)
    ref = a.double() @ w.double().T + bias.double()
    if relu:
        ref = torch.relu(ref)
    out = ops.gemm_nt_bf16(a.to(dev), w.to(dev), bias.to(dev), relu,
                           torch.bfloat16 if out_bf16 else torch.float32).cpu().double()
    tol = 2e-2 if out_bf16 else 1e-4          # bf16 output: half an ulp of the largest values
    assert (out - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_gemm_identity_asymmetric(dev):
    """A = I (padded) with an asymmetric W: catches a transposed C write (cdna guide §3)."""
    from vpr_amd import ops
    K = 64
    a = torch.eye(K).to(torch.bfloat16)
    w = (torch.arange(48 * K).reshape(48, K) % 251).float().to(torch.bfloat16)
    out = ops.gemm_nt_bf16(a.to(dev), w.to(dev)).cpu()
    assert torch.equal(out, w.float().T)


@pytest.mark.parametrize("M,N,K,relu,out_bf16", [
    (256, 256, 128, False, False),     # one tile, the minimum K (two K-tiles)
    (512, 512, 192, True, False),      # odd number of K-tiles
    (1024, 1024, 1024, True, True),    # SALAD layer-1 shape (4 images)
    (300, 260, 256, True, True),       # ragged M and N (clamped rows, masked stores, scalar tail)
    (700, 1024, 4096, False, True),    # long K
    (257, 516, 320, False, False),
])
def test_gemm256(dev, M, N, K, relu, out_bf16):
    """256x256-tile GEMM with LDS-DMA in flight across barriers vs an fp64 reference."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.linspace(-1, 1, N)
    ref = a.double() @ w.double().T + bias.double()
    if relu:
        ref = torch.relu(ref)
    out = ops.gemm_nt_bf16(a.to(dev), w.to(dev), bias.to(dev), relu,
                           torch.bfloat16 if out_bf16 else torch.float32, tile256=True).cpu().double()
    tol = 2e-2 if out_bf16 else 2e-4
    assert (out - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_gemm256_repeatable_under_load(dev):
    """Race screen: the counted-vmcnt pipeline must give bit-identical results run after run (a
    missed wait shows up as a rare wrong tile), here 20 launches of a many-tile problem."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(7)
    a = torch.randn(4096, 1024, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(2048, 1024, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    first = ops.gemm_nt_bf16(a, w, None, False, torch.float32, tile256=True)
    ref = ops.gemm_nt_bf16(a, w, None, False, torch.float32)
    assert (first - ref).abs().max().item() < 1e-3
    for _ in range(20):
        assert torch.equal(ops.gemm_nt_bf16(a, w, None, False, torch.float32, tile256=True), first)


@pytest.mark.parametrize("M,N,K,mode,bias_f32", [(64, 3072, 1024, 0, False), (64, 4096, 1024, 1, False),
                                                 (64, 1024, 4096, 2, False), (5, 40, 96, 0, True),
                                                 (70, 1030, 640, 1, True), (3, 18, 32, 2, False)])
def test_skinny_linear(dev, M, N, K, mode, bias_f32):
    """The cls-row linear layer vs an f64 reference on the same bf16 operands (bias, tanh-GELU, in-place
    accumulate); ragged M / N, more than 64 rows, an output that is a row slice of a wider buffer."""
    from vpr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g)
    b = b if bias_f32 else b.to(torch.bfloat16)
    buf = torch.randn(M + 3, N, generator=g).to(torch.bfloat16)
    acc = a.double() @ w.double().T
    if mode == 0:
        ref = acc + b.double()
    elif mode == 1:
        ref = torch.nn.functional.gelu(acc + b.double(), approximate="tanh")
    else:
        ref = buf[2:2 + M].double() + acc
    dbuf = buf.to(dev)
    ops.skinny_linear_bf16(a.to(dev), w.to(dev), None if mode == 2 else b.to(dev), dbuf[2:2 + M], mode)
    got = dbuf.cpu()
    assert torch.equal(got[:2], buf[:2]) and torch.equal(got[2 + M:], buf[2 + M:])          # neighbours untouched
    err = (got[2:2 + M].double() - ref).abs().max().item()
    assert err < 8e-3 * max(1.0, ref.abs().max().item())                                    # one bf16 rounding
    again = buf.to(dev)
    ops.skinny_linear_bf16(a.to(dev), w.to(dev), None if mode == 2 else b.to(dev), again[2:2 + M], mode)
    assert torch.equal(again.cpu(), got)                                                     # deterministic
