"""rocprofv3 target: similarity + top-k alone.  env: B (default 512), DT = f32|bf16|both (default both), N."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
dev = torch.device("cuda:0")
B, DT, N = int(os.environ.get("B", 512)), os.environ.get("DT", "both"), int(os.environ.get("N", 320000))
D = torch.from_numpy(synth.make_corpus(N, 768)).to(dev)
Db = ops.to_bf16(D) if DT in ("bf16", "both") else None
Qn, _ = synth.make_queries(D[:50000].cpu().numpy(), B); Q = torch.from_numpy(Qn).to(dev)
ws = ops.Workspace(dev)
for _ in range(5):
    if DT in ("bf16", "both"): ops.sim_topk(Q, Db, 100, workspace=ws, exact_on_overflow=False)
    if DT in ("f32", "both"): ops.sim_topk(Q, D, 100, workspace=ws, exact_on_overflow=False)
torch.cuda.synchronize()
