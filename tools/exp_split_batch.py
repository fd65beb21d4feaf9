"""Lab: one C3 batch of 64 queries as ONE call vs as 2 / 4 sub-batches in flight on separate streams (validation_steps)."""
import os, sys, time, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel, GDRRetriever
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base(); N = 320000
sd = synth.make_state_dict(cfg, seed=1234)
names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
D_dev = torch.from_numpy(synth.make_corpus(N, cfg.d_model)).to(dev)
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30))
B, R = 64, 10
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
a_r = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8, kary=30,
                            position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
retr = GDRRetriever(model, D_dev, codec.ClusterIndex(names, offsets, members), a_r)
def timed(fn, reps=7, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
whole = {"source_ids": ids, "source_mask": mask}
print("one call of 64:", timed(lambda: retr.validation_step_i(whole)))
for parts in (2, 4):
    n = B // parts
    subs = [{"source_ids": ids[i * n:(i + 1) * n].contiguous(), "source_mask": mask[i * n:(i + 1) * n].contiguous()} for i in range(parts)]
    print(f"{parts} sub-batches of {n} in flight:", timed(lambda: list(retr.validation_steps(iter(subs), depth=parts))))
