"""rocprofv3 target: docid beam decode alone (encoder + gdr_t5_generate), B=64 beams=10, 4 calls (env B / BEAMS / CALLS; DTYPE=bf16).
Post-process the kernel trace with tools/trace_busy.py to see GPU-busy vs wall span (launch-bound or not)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, R = int(os.environ.get("B", 64)), int(os.environ.get("BEAMS", 10))
cfg = GDRConfig.base()
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
model = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev, ragged=True,
                 dtype=torch.bfloat16 if os.environ.get("DTYPE") == "bf16" else torch.float32,
                 prefix_trie=None if os.environ.get("NO_PREFIX_TABLE") else codec.Trie.from_docids(names, 30))
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
for _ in range(int(os.environ.get("CALLS", 4))):
    model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R,
                   output_scores=True, output_encoder_embedding=os.environ.get("WITH_ENC_EMBEDDING") == "1")   # the retriever path never asks for
                                                                                              # the beam-expanded states (1.9 GB at 512 x 30)
    torch.cuda.synchronize()
