"""A/B of the beam rows' cross-attention forms inside generate(): GDR_ATTN_CROSS_MFMA = (query, head) pairs from which the
MFMA form runs (0 = never, 1 = always).  Prints generate() ms per batch size for the process' setting."""
import json, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234)
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
bf = os.environ.get("DTYPE", "f32") == "bf16"
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30), dtype=torch.bfloat16 if bf else torch.float32)
out = {"setting": os.environ.get("GDR_ATTN_CROSS_MFMA", "default"), "dtype": "bf16" if bf else "f32"}
for s in os.environ.get("SHAPES", "16x10,32x10,64x10,128x10,256x10,512x10,2048x10,64x30,512x30,1x100,8x100").split(","):
    B, R = (int(x) for x in s.split("x"))
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    enc_h, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)
    call = lambda: model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)   # noqa: E731
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[s] = round(sorted(ts)[2] * 1e3, 3)
print(json.dumps(out))
