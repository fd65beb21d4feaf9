import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, _ffi
from gdr_amd._ffi import lib, ptr, stream_ptr, check
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M, N, K = 12308, 2304, 768
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5
Ap, Wp = ops.split_bf16x3(A, padded=False), ops.split_bf16x3(W, padded=False)
C = torch.empty(M, N, device=dev)
for pad in (0, 320, 0, 320, 1024, 320):
    lda = 3 * K + pad
    Ab = torch.zeros(M, lda, dtype=torch.bfloat16, device=dev); Ab[:, :3 * K] = Ap
    Wb = torch.zeros(N, lda, dtype=torch.bfloat16, device=dev); Wb[:, :3 * K] = Wp
    f = lambda: check(lib().gdr_linear_split_bf16(ptr(Ab), lda, ptr(Wb), lda, ptr(C), N, M, N, K, 6, 0, None, None, 0, stream_ptr()), "x")
    print(pad, round(timed(f) * 1e6, 1), "us")
