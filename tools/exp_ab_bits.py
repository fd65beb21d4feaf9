"""Digest of everything the hot path produces on fixed seeded inputs — encoder states, generate() ids + scores at several
shapes (fp32 and bf16), the two-stage result —, to compare two builds of libgdr_hip.so bit for bit on one box:
  python tools/exp_ab_bits.py > a.json;  GDR_HIP_LIB=/path/to/other/libgdr_hip.so python tools/exp_ab_bits.py > b.json
Used for changes that must not move a single bit (cheaper reductions, the three-instruction exact division of the norms)."""
import hashlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel, GDRRetriever
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234)
N = 60000
names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
trie = codec.Trie.from_docids(names, 30)
D = synth.make_corpus(N, 768, seed=5)


def dig(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)).tobytes())
    return h.hexdigest()[:16]


out = {}
for prec, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=trie, dtype=dt)
    for B, R in ((1, 100), (16, 10), (64, 10), (512, 10), (64, 30)):
        ids, mask = synth.make_tokens(B, L=40, seed=11 + B)
        ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        enc_h, pooled = model.enc.forward(ids, mask, want_pooled=True, ragged=False)
        out[f"{prec}_enc_{B}"] = dig(enc_h, pooled)
        enc_r, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)
        res = model.dec.generate(enc_r, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)
        out[f"{prec}_gen_{B}x{R}"] = dig(res[0], res[1], res[2])
        res = model.dec.generate(enc_r, mask, R, 10, 0.8, R)          # every row through adaptor + head (no table)
        out[f"{prec}_gen_notable_{B}x{R}"] = dig(res[0], res[1], res[2])
    if prec == "fp32":
        import types
        a_r = types.SimpleNamespace(num_return_sequences=10, output_vocab_size=30, max_output_length=10, length_penalty=0.8, kary=30,
                                    position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
        ids, mask = synth.make_tokens(64, L=40, seed=3)
        ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        (dec, _), _ = model.generate(ids, attention_mask=mask, max_length=10, num_beams=10, length_penalty=0.8,
                                     num_return_sequences=10, output_scores=True)
        strs = sorted({s for s in codec.decode_token(a_r, dec.cpu().numpy())})[:len(names)]   # the decoded docids name real clusters
        look = codec.ClusterIndex(strs + names[len(strs):], offsets, members)
        retr = GDRRetriever(model, torch.from_numpy(D).to(dev), look, a_r)
        o = retr.validation_step_i({"source_ids": ids, "source_mask": mask})
        out["two_stage_64"] = dig(o["rerank_values"], o["doc_id_tensor"])
    del model
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(out))
