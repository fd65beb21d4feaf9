import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
dev = torch.device("cuda:0")
D = torch.from_numpy(synth.make_corpus(320000, 768)).to(dev); Db = ops.to_bf16(D)
Qn, _ = synth.make_queries(D[:50000].cpu().numpy(), 512); Q = torch.from_numpy(Qn).to(dev)
for _ in range(5): ops.sim_topk(Q, Db, 100); ops.sim_topk(Q, D, 100)
torch.cuda.synchronize()
