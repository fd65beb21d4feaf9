"""generate() wall time per shape, and a digest of its outputs: run once per decode-attention form (one process each) to A/B
the wave-per-(row, head) kernel against the wave-per-(row, 4 heads) kernel on one box.  The digests must be equal."""
import hashlib, json, os, sys, time
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
trie = codec.Trie.from_docids(names, 30)
out = {}
for prec in ("fp32", "bf16"):
    model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=trie, dtype=torch.bfloat16 if prec == "bf16" else torch.float32)
    for B, R in ((16, 10), (64, 10), (256, 10), (512, 10), (2048, 10), (64, 30), (512, 30)):
        ids, mask = synth.make_tokens(B, L=40, seed=11)
        ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        enc_h, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)
        call = lambda: model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)
        for _ in range(2):
            res = call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            res = call()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dec, scores = res[0], res[2]
        h = hashlib.sha1(dec.cpu().numpy().tobytes() + scores.cpu().numpy().tobytes()).hexdigest()[:12]
        out[f"{prec}_{B}x{R}"] = {"ms": round(sorted(ts)[3] * 1e3, 3), "digest": h}
    del model
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(out))
