"""rocprofv3 target: the ragged encoder at the bench batch, fp32 (SPLIT=0) or split-bf16 linears (SPLIT=1), 6 calls."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
enc = ops.T5EncoderHandle(cfg, synth.make_state_dict(cfg, seed=1234, with_decoder=False), dev, split=os.environ.get("SPLIT", "1") == "1")
ids_n, mask_n = synth.make_tokens(512, L=40, seed=11)
ids, mask = torch.from_numpy(ids_n).to(dev), torch.from_numpy(mask_n).to(dev)
for _ in range(6):
    enc.forward(ids, mask, want_hidden=False, ragged=True, live_rows_hint=int(mask_n.sum()))
torch.cuda.synchronize()
