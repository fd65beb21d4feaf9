"""Where a C5 step's time goes (512 queries x 30 beams, bf16, 1M-row corpus): encoder, decode, the device part of stage 2,
the whole step, and streams of steps at pipeline depths 1-3."""
import json, os, sys, time, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, ops, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel, GDRRetriever
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
N = int(os.environ.get("CORPUS", "1000000"))
B, R = int(os.environ.get("B", "512")), int(os.environ.get("R", "30"))
bf = os.environ.get("DTYPE", "bf16") == "bf16"
sd = synth.make_state_dict(cfg, seed=1234)
names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30), dtype=torch.bfloat16 if bf else torch.float32)
D = torch.from_numpy(synth.make_corpus(N, cfg.d_model)).to(dev)
if bf:
    D = ops.to_bf16(D)
ids, mask = synth.make_tokens(B, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
args = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8, kary=30,
                             position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
(dec, _), _ = model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R,
                             output_scores=True)
strs = sorted(set(codec.decode_token(args, dec.cpu().numpy())))[:len(names)]
retr = GDRRetriever(model, D, codec.ClusterIndex(strs + names[len(strs):], offsets, members), args)
batch = {"source_ids": ids, "source_mask": mask}


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e3, 2)


out = {"B": B, "R": R, "corpus": N, "dtype": "bf16" if bf else "f32", "table_levels": model.prefix_table.n_levels}
out["encoder_ms"] = timed(lambda: model.enc.forward(ids, mask, want_pooled=False, ragged=True))
enc_h, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)
out["decode_ms"] = timed(lambda: model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table))
out["step_launch_ms"] = timed(lambda: retr._step_launch(batch))
out["step_ms"] = timed(lambda: retr.validation_step_i(batch))
t0 = time.perf_counter()
st = retr._step_launch(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
o = retr._step_finish(st)
torch.cuda.synchronize()
t3 = time.perf_counter()
out["one_step_split_ms"] = {"enqueue": round((t1 - t0) * 1e3, 2), "gpu_wait": round((t2 - t1) * 1e3, 2), "finish": round((t3 - t2) * 1e3, 2)}
for depth in (1, 2, 3, 4):
    n = 12
    out[f"stream_depth{depth}_ms_per_step"] = round(timed(lambda: list(retr.validation_steps(iter([batch] * n), depth=depth)), reps=3, warm=1) / n, 2)
print(json.dumps(out))
