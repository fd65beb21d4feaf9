"""Does the row stride of the bf16 operands matter to the LDS-DMA linears?  (r06: the split form ran 254 us at a 4 608-byte row stride and
203 us at 5 248.)  gdr_linear_bf16 at the C5 / C2 shapes with A (and W) rows padded by `pad` elements."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gdr_amd import ops
from gdr_amd._ffi import lib, ptr, stream_ptr, check
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


for M, N, K in ((15360, 2304, 768), (15360, 768, 768), (15360, 3072, 768), (15360, 768, 3072), (12308, 2304, 768), (12308, 768, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    C = torch.empty(M, N, device=dev)
    line = f"M {M} N {N} K {K}:"
    for pa, pw in ((0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (192, 192), (320, 320)):
        Ab = torch.zeros(M, K + pa, dtype=torch.bfloat16, device=dev); Ab[:, :K] = A
        Wb = torch.zeros(N, K + pw, dtype=torch.bfloat16, device=dev); Wb[:, :K] = W
        f = lambda: check(lib().gdr_linear_bf16(ptr(Ab), K + pa, ptr(Wb), K + pw, ptr(C), N, M, N, K, 0, None, None, 0, stream_ptr()), "x")
        line += f"  pad A{pa}/W{pw} {timed(f) * 1e6:6.1f}"
    print(line)
