#!/usr/bin/env python3
"""C5's decode leg (64 queries x 30 beams, prefix table) in fp32 and in the bf16 precision mode: generate() ms and the kernel
mix of the bf16 call (rocprofv3 target when PROF=1)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234)
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
trie = codec.Trie.from_docids(names, 30)
B, R = int(os.environ.get("B", 64)), int(os.environ.get("BEAMS", 30))
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
for label, dt in (("bf16", torch.bfloat16),) if os.environ.get("PROF") == "1" or os.environ.get("ONLY") == "bf16" else (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=trie, dtype=dt)
    g = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R, output_scores=True)
    for _ in range(2):
        g()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, "generate ms", round(sorted(ts)[2] * 1e3, 2), flush=True)
    del model
    torch.cuda.empty_cache()
