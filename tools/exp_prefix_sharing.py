#!/usr/bin/env python3
"""How much do the R beams of a query share their ancestors?  Rows with the same token prefix of length p + 1 read the same K / V
row at position p in the decode self-attention: per position, the number of DISTINCT prefixes among a query's final beams (mean
over queries), for the bench's synthetic weights.  env: B, BEAMS, DTYPE."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, R = int(os.environ.get("B", 64)), int(os.environ.get("BEAMS", 30))
cfg = GDRConfig.base()
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
model = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev, ragged=True,
                 dtype=torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32,
                 prefix_trie=codec.Trie.from_docids(names, 30))
ids, mask = synth.make_tokens(B, L=40, seed=11)
(dec, _), _ = model.generate(torch.from_numpy(ids).to(dev), attention_mask=torch.from_numpy(mask).to(dev), max_length=10, num_beams=R,
                             length_penalty=0.8, num_return_sequences=R, output_scores=True)
d = dec.cpu().numpy().reshape(B, R, -1)
T = d.shape[2]
out = []
for p in range(T):
    out.append(float(np.mean([len({tuple(row[:p + 1]) for row in d[b]}) for b in range(B)])))
print("distinct prefixes per position (of %d beams):" % R, [round(x, 1) for x in out])
print("gathered rows per query: %d, distinct: %.1f -> sharing factor %.2f" % (R * (T - 1), sum(out[:T - 1]), R * (T - 1) / sum(out[:T - 1])))
