"""bf16 linear (gdr_linear_bf16) at the encoder's / decode's call shapes: us and TFLOP/s per shape (HIP-event timed, 30 calls)."""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, _ffi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
out = {"setting": os.environ.get("GDR_LAB_BF16_BM64_BELOW", "default")}
for M in (int(x) for x in os.environ.get("MS", "1920,4096,12308,15360,20480").split(",")):
    for name, N, K, res in (("qkv", 2304, 768, False), ("o", 768, 768, True), ("wi", 3072, 768, False), ("wo", 768, 3072, True),
                            ("lin1", 2048, 768, False), ("lin2", 768, 2048, True)) + \
            ((("head", 23808, 768, False),) if os.environ.get("HEAD") else ()) + \
            ((("o_nores", 768, 768, False), ("wo_nores", 768, 3072, False)) if os.environ.get("NORES") else ()):
        a = ops.to_bf16((torch.randn(M, K, generator=g) * 0.05).to(dev))
        w = ops.to_bf16((torch.randn(N, K, generator=g) * 0.05).to(dev))
        r = (torch.randn(M, N, generator=g) * 0.05).to(dev) if res else None
        o = torch.empty(M, N, device=dev)
        epi = _ffi.EPI_RESIDUAL if res else _ffi.EPI_NONE
        for _ in range(5):
            ops.linear_bf16(a, w, epilogue=epi, residual=r, out=o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            ops.linear_bf16(a, w, epilogue=epi, residual=r, out=o)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        out[f"M{M}_{name}"] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
print(json.dumps(out))
