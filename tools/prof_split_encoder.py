"""rocprofv3 target: the ragged encoder at the bench batch; SPLIT = 0 (fp32), 2 (fp16 x 2), 6 / 3 (bf16 x 3 forms); 6 calls."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
enc = ops.T5EncoderHandle(cfg, synth.make_state_dict(cfg, seed=1234, with_decoder=False), dev, split=int(os.environ.get("SPLIT", "2")))
ids_n, mask_n = synth.make_tokens(512, L=40, seed=11)
ids, mask = torch.from_numpy(ids_n).to(dev), torch.from_numpy(mask_n).to(dev)
for _ in range(6):
    enc.forward(ids, mask, want_hidden=False, ragged=True, live_rows_hint=int(mask_n.sum()))
torch.cuda.synchronize()
