#!/usr/bin/env python3
"""rocprofv3 target: the packed T5-base encoder pass of a small batch (C3's 64 queries x 40 tokens by default), 20 calls."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import synth, ops
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234)
enc = ops.T5EncoderHandle(cfg, sd, dev)
B = int(os.environ.get("B", "64"))
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
ragged = os.environ.get("RAGGED", "1") == "1"
for _ in range(3):
    enc.forward(ids, mask, want_pooled=False, ragged=ragged)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    enc.forward(ids, mask, want_pooled=False, ragged=ragged)
e1.record()
torch.cuda.synchronize()
print(f"B={B} ragged={ragged} live rows={int(mask.sum())}: {e0.elapsed_time(e1) / 20:.3f} ms per encoder pass")
