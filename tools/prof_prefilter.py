"""rocprofv3 target: ops.sim_topk through the bf16 pre-filter at B (env, default 32), 5 calls."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
Dn = synth.make_corpus(320000, 768)
D = torch.from_numpy(Dn).to(dev)
P = ops.PrefilteredCorpus(D)
Qn, _ = synth.make_queries(Dn[:50000], B); Q = torch.from_numpy(Qn).to(dev)
ws = ops.Workspace(dev)
for _ in range(5):
    ops.sim_topk(Q, P, 100, workspace=ws, exact_on_overflow=False)
torch.cuda.synchronize()
