import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
enc = ops.T5EncoderHandle(cfg, sd, dev)
ids, mask = synth.make_tokens(int(os.environ.get("EXP_B", "512")), L=40, seed=11)
it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
print("live rows", int(mask.sum()), "of", mask.size)
ragged = os.environ.get("EXP_PADDED", "0") != "1"
for _ in range(13):
    if ragged:
        enc.forward(it, mt, want_hidden=False, ragged=True, live_rows_hint=int(mask.sum()))
    else:
        enc.forward(it, mt)
torch.cuda.synchronize()
