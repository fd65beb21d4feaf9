"""rocprofv3 target: one linear shape, 6 launches.  env: N, K, M (default FFN2 of the bench: 768, 3072, 20480)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", 20480)), int(os.environ.get("N", 768)), int(os.environ.get("K", 3072))
a = (torch.randn(M, K) * 0.05).to(dev); w = (torch.randn(N, K) * 0.05).to(dev); out = torch.empty(M, N, device=dev)
for _ in range(6): ops.linear(a, w, out=out)
torch.cuda.synchronize()
