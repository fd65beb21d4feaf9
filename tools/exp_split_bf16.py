#!/usr/bin/env python3
"""Exploratory: a split-bf16 linear — A and W each as hi + mid + lo bf16 planes (24 mantissa bits carried), the six
leading products hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid on the bf16 MFMA path with fp32 accumulate — against the strict-fp32
MFMA linear, on the encoder's four shapes.  Measured WITHOUT a new kernel: the six products are ONE bf16 GEMM over a 6x longer
contraction, A' = [hi | hi | mid | hi | lo | mid], W' = [hi | mid | hi | lo | hi | mid] (so the numbers are what the existing bf16 kernels
give a K' = 6 K problem; a dedicated kernel would hold the three planes of a k-tile in LDS once and save the duplicated operand bytes).
Prints per shape: error of both forms against float64 on sampled rows, time and fp32-equivalent TFLOP/s (2 M N K / t)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
g = torch.Generator(device="cpu").manual_seed(5)
for M in (12308, 20480):
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
        ah, am, al = split3(A)
        wh, wm, wl = split3(W)
        A6 = torch.cat([ah, ah, am, ah, al, am], 1).contiguous()
        W6 = torch.cat([wh, wm, wh, wl, wh, wm], 1).contiguous()
        A3 = torch.cat([ah, ah, am], 1).contiguous()
        W3 = torch.cat([wh, wm, wh], 1).contiguous()
        C = torch.empty(M, N, device=dev)
        c32 = ops.linear(A, W, splitk_ws=ws).clone()
        c6 = ops.linear_bf16(A6, W6).clone()
        c3 = ops.linear_bf16(A3, W3).clone()
        c1 = ops.linear_bf16(ah, wh).clone()
        Ap, Wp = ops.split_bf16x3(A), ops.split_bf16x3(W)                 # the product form: three planes per (padded) row, 6 B / element
        assert torch.equal(Ap[:, :K], ah) and torch.equal(Ap[:, K:2 * K], am) and torch.equal(Ap[:, 2 * K:3 * K], al)
        cp = ops.linear_split_bf16(Ap, Wp, K).clone()
        assert torch.equal(cp, c6), "plane-addressed kernel differs from the concatenated-operand GEMM"
        tp = timed(lambda: ops.linear_split_bf16(Ap, Wp, K, out=C))
        tsplit = timed(lambda: ops.split_bf16x3(A))
        cp3 = ops.linear_split_bf16(Ap, Wp, K, terms=3).clone()
        assert torch.equal(cp3, c3), "3-term plane-addressed kernel differs from the concatenated-operand GEMM"
        tp3 = timed(lambda: ops.linear_split_bf16(Ap, Wp, K, out=C, terms=3))
        Ah, Wh = ops.split_f16x2(A), ops.split_f16x2(W)                   # fp16 x 2: [hi | (x - hi) * 2^11], 22 bits, three blocks
        ch = ops.linear_split_bf16(Ah, Wh, K, terms=2).clone()
        th = timed(lambda: ops.linear_split_bf16(Ah, Wh, K, out=C, terms=2))
        rows = torch.arange(0, M, max(1, M // 64), device=dev)[:64]
        ref = (A[rows].double() @ W.double().T)
        scale = float(ref.abs().mean())
        err = lambda c: float((c[rows].double() - ref).abs().max()) / scale     # noqa: E731
        t32 = timed(lambda: ops.linear(A, W, out=C, splitk_ws=ws))
        t6 = timed(lambda: ops.linear_bf16(A6, W6, out=C))
        t3 = timed(lambda: ops.linear_bf16(A3, W3, out=C))
        fl = 2.0 * M * N * K
        print(f"M {M:6d} N {N:5d} K {K:5d} | max err / mean|c|: fp32 {err(c32):.2e}  split6 {err(c6):.2e}  split3 {err(c3):.2e}  bf16 {err(c1):.2e}"
              f" | fp32 {t32 * 1e6:7.1f} us {fl / t32 / 1e12:6.1f} TF  split6 {t6 * 1e6:7.1f} us {fl / t6 / 1e12:6.1f} TF-equiv ({t32 / t6:4.2f}x)"
              f"  split3 {t3 * 1e6:7.1f} us ({t32 / t3:4.2f}x)  [form {ops.lib().gdr_linear_bf16_tile_form(M, N, 6 * K, 0)}]"
              f"  | planes kernel {tp * 1e6:7.1f} us ({t32 / tp:4.2f}x), 3 terms {tp3 * 1e6:7.1f} us ({t32 / tp3:4.2f}x), fp16x2 {th * 1e6:7.1f} us ({t32 / th:4.2f}x) err {err(ch):.2e}, split of A {tsplit * 1e6:6.1f} us")
