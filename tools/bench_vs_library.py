#!/usr/bin/env python3
"""Evidence, not product: the hand-written linear kernels against the vendor library on the encoder's GEMM shapes
(torch.matmul -> rocBLAS / hipBLASLt on this image).  fp32 and bf16, padded (20 480 rows) and ragged (12 308 rows) batch."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


rows = []
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)   # split-K / stream-K scratch, as the encoder's linears have it
for M in (20480, 15360, 12308):
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        C = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        t_ours = timed(lambda: ops.linear(A, W, out=C, splitk_ws=ws))
        t_lib = timed(lambda: torch.matmul(A, W.t(), out=C))
        Ab, Wb = A.to(torch.bfloat16), W.to(torch.bfloat16)
        t_ours16 = timed(lambda: ops.linear_bf16(Ab, Wb, out=C))
        Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_lib16 = timed(lambda: torch.matmul(Ab, Wb.t(), out=Cb))
        r = dict(M=M, N=N, K=K, f32_ours_tflops=round(fl / t_ours / 1e12, 1), f32_library_tflops=round(fl / t_lib / 1e12, 1),
                 bf16_ours_tflops=round(fl / t_ours16 / 1e12, 1), bf16_library_tflops=round(fl / t_lib16 / 1e12, 1))
        rows.append(r)
        print(r)
