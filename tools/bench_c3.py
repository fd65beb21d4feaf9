"""Quick C3 numbers on one GPU (a subset of bench.py's `stages`, minutes faster): generate() / decode at 64 x 10 beams and
1 x 100 beams, the two-stage step start to finish, and the stage-2 part alone (device cluster lookup + rerank)."""
import json, os, sys, time, types
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth, ops
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel, GDRRetriever
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


cfg = GDRConfig.base()
N = 320000
sd = synth.make_state_dict(cfg, seed=1234)
names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
D_dev = torch.from_numpy(synth.make_corpus(N, cfg.d_model)).to(dev)
_trie = codec.Trie.from_docids(names, 30)
# TRIE=1: trie-constrained beams (every hypothesis is a corpus docid; all beams end two steps after the deepest leaf)
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=_trie, trie=_trie if os.environ.get("TRIE") == "1" else None)
out = {}
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "64x10,1x100").split(",")]
for B, R in shapes:
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    g = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8,   # noqa: E731
                               num_return_sequences=R, output_scores=True, output_encoder_embedding=True)
    t = timed(g)
    t_enc = timed(lambda: model.enc.forward(ids, mask, want_pooled=False, ragged=True), reps=5, warm=1)
    a_r = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8, kary=30,
                                position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
    (dec, _), _ = g()
    strs = sorted({s for s in codec.decode_token(a_r, dec.cpu().numpy())})[:len(names)]
    look = codec.ClusterIndex(strs + names[len(strs):], offsets, members)
    retr = GDRRetriever(model, D_dev, look, a_r)
    batch = {"source_ids": ids, "source_mask": mask}
    t3 = timed(lambda: retr.validation_step_i(batch))
    retr_h = GDRRetriever(model, D_dev, look, a_r, device_candidates=False)
    t3h = timed(lambda: retr_h.validation_step_i(batch))
    # stage 2 alone on the device: cluster lookup + rerank
    st = retr._step_launch(batch)
    torch.cuda.synchronize()
    dci = retr._device_index()
    q = st["enc_h"][:, 0].contiguous()
    bs = st["scores"].to(torch.float32).view(B, R)

    def stage2():
        _cl, offs, cids, stride = dci.candidates(st["ids"], B, R)
        return ops.rerank_topk(q, D_dev, offs, cids, bs, a_r.score_rate, R, max_cand=stride, cand_stride=stride)

    t2 = timed(stage2, reps=20, warm=3)
    _cl, offs, cids, stride = dci.candidates(st["ids"], B, R)
    ncand = int(offs[:, R].sum().item())
    out[f"B{B}_beam{R}"] = {"generate_ms": t, "encoder_ms": t_enc, "decode_ms": t - t_enc, "two_stage_ms": t3,
                            "two_stage_qps": B / t3 * 1e3, "two_stage_host_candidates_ms": t3h, "stage2_device_ms": t2,
                            "candidates": ncand, "stage2_gather_gbs": ncand * cfg.d_model * 4 / (t2 * 1e-3) / 1e9}
from gdr_amd import _ffi
out["early_exits"] = int(_ffi.lib().gdr_t5_generate_early_exits())
print(json.dumps(out, indent=1))
