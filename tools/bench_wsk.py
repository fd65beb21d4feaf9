#!/usr/bin/env python3
"""The decode linears at M rows: wave-split-K kernel (gemm_wsk.hip) against the round-2 split-K path (64x64 tiles + reduce),
wall time per call over back-to-back launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, _ffi
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
shapes = [("o / q_c / o_c", 768, 768), ("qkv", 2304, 768), ("wi", 3072, 768), ("wo", 768, 3072)]


def timeit(fn, n=100):
    """GPU time per call: n calls captured into one HIP graph (no host launch cost between them), replayed."""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(ts)[2]


for M in [int(x) for x in os.environ.get("MS", "640,100,64,1920").split(",")]:
    for name, N, K in shapes:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        C = torch.empty(M, N, device=dev)
        R = torch.randn(M, N, device=dev)
        part = torch.rand(M, K // 64, device=dev) * 64
        lnw = torch.ones(K, device=dev)
        t_old = timeit(lambda: ops.linear(A, W, out=C, splitk_ws=ws))
        t_new = timeit(lambda: ops.linear_wsk(A, W, out=C))
        t_new_n = timeit(lambda: ops.linear_wsk(A, W, out=C, part_in=part, norm_w=lnw)) if K == 768 else float("nan")
        t_new_r = timeit(lambda: ops.linear_wsk(A, W, out=C, residual=R, want_part=True))   # incl. a caching-allocator hit per call
        fl = 2.0 * M * N * K
        print(f"M={M:5d} {name:14s} N={N:5d} K={K:5d}  old {t_old:6.1f} us ({fl / t_old / 1e6:5.1f} TF)   wsk {t_new:6.1f} us "
              f"({fl / t_new / 1e6:5.1f} TF)   wsk+norm {t_new_n:6.1f}   wsk+res+part {t_new_r:6.1f}")
