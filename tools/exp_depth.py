"""Lab: C3 stream of batches with 1..4 in flight (GDRRetriever.validation_steps depth)."""
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
graph = bool(int(os.environ.get("GRAPH", "0")))
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30), graph=graph)
B, R = int(os.environ.get("B", 64)), int(os.environ.get("BEAMS", 10))
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
a_r = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8, kary=30,
                            position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
retr = GDRRetriever(model, D_dev, codec.ClusterIndex(names, offsets, members), a_r)
batch = {"source_ids": ids, "source_mask": mask}
nb = 12
for d_ in (1, 2, 3, 4):
    list(retr.validation_steps(iter([batch] * 4), depth=d_))
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); list(retr.validation_steps(iter([batch] * nb), depth=d_)); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / nb * 1e3)
    print(f"graph={graph} depth {d_}: {sorted(ts)[1]:.2f} ms per batch = {B / sorted(ts)[1] * 1e3:.0f} q/s")
