"""The decode chain's wave-split-K linear (csrc/gemm_wsk.hip; the linears of transformers/modeling_t5.py:360-364,413,182-185
at M = batch x beams rows) on its own: plain, ReLU and residual epilogues, the T5LayerNorm (modeling_t5.py:164-171) folded
into the A operand from per-tile sums of squares, the sums of squares it emits for the next linear, edge rows — against a
float64 reference — and a chain of two linears with the norm in between against the unfused form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rms(x, w, eps):
    x = x.double()
    return (x / torch.sqrt((x * x).mean(-1, keepdim=True) + eps) * w.double())


@pytest.mark.parametrize("M,N,K", [(640, 768, 768), (640, 2304, 768), (640, 3072, 768), (640, 768, 3072), (100, 768, 768),
                                   (1, 768, 768), (33, 128, 128), (1920, 768, 768), (5120, 3072, 768), (77, 64, 256)])
def test_wsk_linear_epilogues_vs_float64(dev, M, N, K):
    from gdr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    R = torch.randn(M, N, generator=g)
    Ad, Wd, Rd = A.to(dev), W.to(dev), R.to(dev)
    ref = A.double() @ W.double().T
    tol = dict(rtol=2e-5, atol=2e-5 * K ** 0.5)
    torch.testing.assert_close(ops.linear_wsk(Ad, Wd).cpu().double(), ref, **tol)
    torch.testing.assert_close(ops.linear_wsk(Ad, Wd, relu=True).cpu().double(), ref.clamp(min=0), **tol)
    out, part = ops.linear_wsk(Ad, Wd, residual=Rd, want_part=True)
    want = ref + R.double()
    torch.testing.assert_close(out.cpu().double(), want, **tol)
    torch.testing.assert_close(part.cpu().double(), (want ** 2).view(M, N // 64, 64).sum(-1), rtol=1e-4, atol=1e-4)
    x = Rd.clone()                                                          # in place, as the residual stream runs
    ops.linear_wsk(Ad, Wd, residual=x, out=x)
    assert torch.equal(x, out)
    assert torch.equal(ops.linear_wsk(Ad, Wd), ops.linear_wsk(Ad, Wd))      # deterministic


@pytest.mark.parametrize("M", [640, 100, 7])
def test_wsk_fused_rmsnorm_chain_vs_unfused(dev, M):
    """y = relu(T5LayerNorm(x + ctx Wo^T) Wi^T): the producer emits the row sums of squares, the consumer folds the norm
    into its A operand — against float64 and against the unfused kernels (rmsnorm launch in between) at fp32 tolerance."""
    from gdr_amd import ops
    d, dff, eps = 768, 3072, 1e-6
    g = torch.Generator().manual_seed(M)
    ctx, x = torch.randn(M, d, generator=g), torch.randn(M, d, generator=g) * 3
    Wo, Wi = torch.randn(d, d, generator=g) * d ** -0.5, torch.randn(dff, d, generator=g) * d ** -0.5
    lnw = 1 + 0.1 * torch.randn(d, generator=g)
    h, part = ops.linear_wsk(ctx.to(dev), Wo.to(dev), residual=x.to(dev), want_part=True)
    y = ops.linear_wsk(h, Wi.to(dev), relu=True, part_in=part, norm_w=lnw.to(dev), eps=eps)
    h64 = x.double() + ctx.double() @ Wo.double().T
    y64 = (_rms(h64, lnw, eps) @ Wi.double().T).clamp(min=0)
    torch.testing.assert_close(h.cpu().double(), h64, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(y.cpu().double(), y64, rtol=1e-4, atol=2e-4)


def test_wsk_rejects_unserved_shapes(dev):
    from gdr_amd import _ffi, ops
    A, W = torch.zeros(8, 96, device=dev), torch.zeros(64, 96, device=dev)
    with pytest.raises(_ffi.GdrError, match="not served"):
        ops.linear_wsk(A, W)
    with pytest.raises(_ffi.GdrError, match="not served"):
        ops.linear_wsk(torch.zeros(8, 128, device=dev), torch.zeros(40, 128, device=dev))
