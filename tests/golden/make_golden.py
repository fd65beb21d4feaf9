#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference on CPU).

Build-container only: /root/reference does not exist on the GPU box and nothing at test/bench time
reads it.  Only the arrays written here (inputs + the reference's outputs) are committed; no
reference source is copied.  Import shims follow SURVEY.md Appendix D (they bypass the vendored
transformers/__init__.py, which needs sacremoses/protobuf<4, and stub pytorch_lightning).

Weights are this repo's seeded synthetic state_dicts (gdr_amd/synth.py) loaded into the reference
classes with load_state_dict, so the fixtures pin "reference code + these weights + these inputs".

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only NAME ...]
"""
import argparse
import builtins
import importlib
import io
import contextlib
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/GDR_model"
sys.dont_write_bytecode = True

from gdr_amd.config import GDRConfig            # noqa: E402
from gdr_amd import synth                        # noqa: E402


# ------------------------------------------------------------------------------------ shims
def import_reference():
    pkg = types.ModuleType("transformers")
    pkg.__path__ = [REF + "/transformers"]
    pkg.__version__ = "3.4.0"
    sys.modules["transformers"] = pkg

    def grab(mod, names):
        m = importlib.import_module("transformers." + mod)
        for n in names:
            setattr(pkg, n, getattr(m, n))

    grab("modeling_t5", ["T5ForConditionalGeneration"])
    grab("configuration_t5", ["T5Config"])
    grab("tokenization_t5", ["T5Tokenizer"])
    grab("optimization", ["AdamW", "get_linear_schedule_with_warmup"])
    grab("modeling_bert", ["BertModel"])
    grab("configuration_bert", ["BertConfig"])
    grab("tokenization_bert", ["BertTokenizer"])
    grab("modeling_dpr", ["DPRQuestionEncoder", "DPRContextEncoder"])
    grab("tokenization_dpr", ["DPRQuestionEncoderTokenizer"])
    grab("modeling_utils", ["PreTrainedModel"])
    pkg.AutoModel = object
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.__version__ = "stub"
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    import main_models  # noqa
    import main_utils   # noqa
    import main_metrics  # noqa
    return pkg, main_models, main_utils, main_metrics


def ref_t5(cfg: GDRConfig, sd):
    """Instantiate the reference T5ForConditionalGeneration with GDR's kwargs (main_models.py:748-780)."""
    from transformers.configuration_t5 import T5Config
    from transformers.modeling_t5 import T5ForConditionalGeneration
    c = T5Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
                 num_layers=cfg.num_layers, num_decoder_layers=cfg.num_decoder_layers,
                 num_heads=cfg.num_heads, relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                 dropout_rate=0.1, layer_norm_epsilon=cfg.layer_norm_epsilon,
                 decode_embedding=2, decode_vocab_size=cfg.decode_vocab_size,
                 output_vocab_size=cfg.output_vocab_size, adaptor_decode=1, adaptor_efficient=1,
                 adaptor_layer_num=cfg.adaptor_layer_num, tie_decode_embedding=1, tie_word_embeddings=0,
                 Rdrop=0.1, Rdrop_loss="KL", Rdrop_only_decoder=0, denoising=0, multiple_decoder=0,
                 decoder_num=1, max_output_length=cfg.max_output_length, embedding_distillation=0.0,
                 weight_distillation=0.0, decoder_start_token_id=0, hierarchic_decode=0,
                 pad_token_id=0, eos_token_id=1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = T5ForConditionalGeneration(c)
    if cfg.adaptor_ff != 2048:
        # torch's TransformerDecoderLayer default dim_feedforward is 2048; the tiny fixture shrinks it
        layer = torch.nn.TransformerDecoderLayer(d_model=cfg.d_model, nhead=cfg.adaptor_nhead,
                                                 dim_feedforward=cfg.adaptor_ff)
        m.adaptor = torch.nn.TransformerDecoder(layer, num_layers=cfg.adaptor_layer_num)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("denoising" in k for k in missing), missing
    return m.eval()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------ fixtures
def g_buckets():
    """G2: _relative_position_bucket for all (i,j) in [0,128)^2, both modes (modeling_t5.py:242-288)."""
    from transformers.modeling_t5 import T5Attention
    ctx = torch.arange(128)[:, None]
    mem = torch.arange(128)[None, :]
    rel = mem - ctx
    bi = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32)
    uni = T5Attention._relative_position_bucket(rel, bidirectional=False, num_buckets=32)
    save("g2_buckets", bidirectional=bi.to(torch.int8), unidirectional=uni.to(torch.int8))


def g_encoder_tiny():
    """G1a: tiny encoder, ragged padding: last_hidden_state + layer-0 position_bias (+mask)."""
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=1234)
    m = ref_t5(cfg, sd)
    ids, mask = synth.make_tokens(6, L=8, vocab_hi=cfg.vocab_size, seed=3, min_len=2)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    with torch.no_grad():
        out = m.get_encoder()(ids_t, attention_mask=mask_t, return_dict=True, output_attentions=False)
        # position bias of block 0 (with the additive pad mask folded in, modeling_t5.py:399-400)
        emb = m.shared(ids_t)
        ext = m.encoder.get_extended_attention_mask(mask_t, ids_t.shape, ids_t.device)
        blk0 = m.encoder.block[0].layer[0]
        _, _, pb = blk0(emb, attention_mask=ext)
    save("g1_encoder_tiny", input_ids=ids, attention_mask=mask, last_hidden_state=out.last_hidden_state,
         position_bias=pb, seed=1234)


def g_encoder_base():
    """G1b: t5-base-shaped encoder, B=4, L=40: pooled CLS rows + 8 sampled token rows."""
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
    from transformers.configuration_t5 import T5Config
    from transformers.modeling_t5 import T5Stack
    c = T5Config(vocab_size=cfg.vocab_size, d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12,
                 dropout_rate=0.1, is_decoder=False, use_cache=False, is_encoder_decoder=False)
    shared = torch.nn.Embedding(cfg.vocab_size, 768)
    enc = T5Stack(c, shared)
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc.load_state_dict(esd)
    enc.eval()
    ids, mask = synth.make_tokens(4, L=40, seed=11)
    with torch.no_grad():
        h = enc(torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), return_dict=True).last_hidden_state
    rows = np.array([[0, 1], [0, 7], [1, 5], [1, 39], [2, 2], [2, 20], [3, 3], [3, 30]])
    save("g1_encoder_base", input_ids=ids, attention_mask=mask, pooled=h[:, 0],
         sample_rc=rows, sample_rows=h[rows[:, 0], rows[:, 1]], seed=1234)


def g_sim_topk():
    """G3: dense.py `compute_similarity` (q @ p.T) + topk(10) on C1 shape 128 x 1000 x 768."""
    tmp = tempfile.mkdtemp()
    try:
        os.makedirs(tmp + "/refdense")
        open(tmp + "/refdense/__init__.py", "w").close()
        for f in ("dense.py", "encoder.py"):
            shutil.copy(REF + "/" + f, tmp + "/refdense/" + f)      # temp dir only, never the repo
        builtins.ModelArguments = builtins.TrainingArguments = object
        # encoder.py needs only PreTrainedModel / AutoModel / file_utils.ModelOutput, all of which the
        # stub `transformers` package of import_reference() already resolves to the vendored copy
        sys.path.insert(0, tmp)
        try:
            dense = importlib.import_module("refdense.dense")
        finally:
            sys.path.remove(tmp)
        D = synth.make_corpus(1000, 768)
        Q, gold = synth.make_queries(D, 128)
        S = dense.DenseModel.compute_similarity(None, torch.from_numpy(Q), torch.from_numpy(D))
        v, i = S.topk(10, dim=1, largest=True, sorted=True)
        # DensePooler contract (dense.py:18-27) on a few hidden states
        torch.manual_seed(20240905)                              # nn.Linear's init draws from the global generator
        pool = dense.DensePooler(48, 32, normalize=True)
        g = torch.Generator().manual_seed(5)
        hid = torch.randn(3, 4, 48, generator=g)
        with torch.no_grad():
            pq = pool(q=hid)
        save("g3_sim_topk", values=v, indices=i.to(torch.int32), gold=gold,
             pool_hidden=hid, pool_w=pool.linear_q.weight, pool_b=pool.linear_q.bias, pool_out=pq)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def g_dense_model():
    """G13: dense.DenseModel end to end (dense.py:30-54, encoder.py:77-113 eval branch) over the reference T5 encoder
    (tiny shape): encode_query / encode_passage with a DensePooler (separate q / p linears, L2 normalise) and without,
    forward(query=, passage=).scores, and topk over the scores."""
    tmp = tempfile.mkdtemp()
    try:
        os.makedirs(tmp + "/refdense2")
        open(tmp + "/refdense2/__init__.py", "w").close()
        for f in ("dense.py", "encoder.py"):
            shutil.copy(REF + "/" + f, tmp + "/refdense2/" + f)     # temp dir only, never the repo
        builtins.ModelArguments = builtins.TrainingArguments = object
        sys.path.insert(0, tmp)
        try:
            dense = importlib.import_module("refdense2.dense")
        finally:
            sys.path.remove(tmp)
        cfg = GDRConfig.tiny()
        sd = synth.make_state_dict(cfg, seed=1234)
        enc = ref_t5(cfg, sd).get_encoder()
        qi, qm = synth.make_tokens(5, L=8, vocab_hi=cfg.vocab_size, seed=21, min_len=2)
        pi, pm = synth.make_tokens(9, L=12, vocab_hi=cfg.vocab_size, seed=22, min_len=3)
        qry = {"input_ids": torch.from_numpy(qi), "attention_mask": torch.from_numpy(qm)}
        psg = {"input_ids": torch.from_numpy(pi), "attention_mask": torch.from_numpy(pm)}
        torch.manual_seed(9)
        pool = dense.DensePooler(cfg.d_model, 32, normalize=True)
        out = {}
        with torch.no_grad():
            for name, pooler in (("pool", pool), ("cls", None)):
                m = dense.DenseModel(enc, enc, pooler=pooler).eval()
                o = m(query=qry, passage=psg)
                assert o.loss is None
                assert torch.equal(o.q_reps, m.encode_query(qry)) and torch.equal(o.p_reps, m.encode_passage(psg))
                v, i = o.scores.topk(3, dim=1, largest=True, sorted=True)
                out.update({f"{name}_q_reps": o.q_reps, f"{name}_p_reps": o.p_reps, f"{name}_scores": o.scores,
                            f"{name}_top_v": v, f"{name}_top_i": i.to(torch.int32)})
        save("g13_dense_model", seed=1234, q_ids=qi, q_mask=qm, p_ids=pi, p_mask=pm,
             wq=pool.linear_q.weight, bq=pool.linear_q.bias, wp=pool.linear_p.weight, bp=pool.linear_p.bias, **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def g_decode_logits_tiny():
    """G8: T5ForConditionalGeneration.forward decode branch (decoder + adaptor head + positional mask), tiny."""
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=1234)
    m = ref_t5(cfg, sd)
    ids, mask = synth.make_tokens(3, L=8, vocab_hi=cfg.vocab_size, seed=3, min_len=2)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    V = cfg.output_vocab_size
    g = np.random.Generator(np.random.PCG64(9))
    t = 4
    dec = np.zeros((3, t), dtype=np.int64)
    for p in range(1, t):
        dec[:, p] = (p - 1) * V + 2 + g.integers(0, V, size=3)
    with torch.no_grad():
        enc = m.get_encoder()(ids_t, attention_mask=mask_t, return_dict=True)
        out = m(decoder_input_ids=torch.from_numpy(dec), encoder_outputs=enc, attention_mask=mask_t,
                use_cache=False, return_dict=True)
    save("g8_decode_logits_tiny", input_ids=ids, attention_mask=mask, decoder_input_ids=dec,
         logits=out.logits, enc=enc.last_hidden_state, seed=1234)


def _generate(m, ids_t, mask_t, cfg, R, lp=0.8):
    steps = []
    orig = torch.topk

    def spy(x, k, *a, **kw):
        r = orig(x, k, *a, **kw)
        if x.dim() == 2 and k == 2 * R:
            steps.append((r[0].clone(), r[1].clone()))
        return r

    torch.topk = spy
    try:
        with torch.no_grad():
            (outs, scores), enc = quiet(
                m.generate, ids_t, attention_mask=mask_t, use_cache=False, max_length=cfg.max_output_length,
                num_beams=R, length_penalty=lp, num_return_sequences=R, early_stopping=False,
                decode_embedding=2, decode_vocab_size=cfg.decode_vocab_size, decode_tree=None, decoder_index=-1,
                output_scores=True, output_encoder_embedding=True, cluster_constraint=None)
    finally:
        torch.topk = orig
    return outs, scores, enc.last_hidden_state, steps


def g_generate_tiny():
    """G5a: full generate(), tiny config, beam 4, B=3, with the per-step top-2R trace."""
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=1234)
    m = ref_t5(cfg, sd)
    ids, mask = synth.make_tokens(3, L=8, vocab_hi=cfg.vocab_size, seed=3, min_len=2)
    outs, scores, enc, steps = _generate(m, torch.from_numpy(ids), torch.from_numpy(mask), cfg, R=4)
    save("g5_generate_tiny", input_ids=ids, attention_mask=mask, decoded=outs, scores=np.array(scores, np.float64),
         enc=enc, step_scores=torch.stack([s for s, _ in steps]), step_tokens=torch.stack([t for _, t in steps]),
         num_beams=4, length_penalty=0.8, seed=1234)


def g_generate_base():
    """G5b: full generate(), t5-base shape, beam 10, B=2 (reference formulation, use_cache=False)."""
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234)
    m = ref_t5(cfg, sd)
    ids, mask = synth.make_tokens(2, L=40, seed=11)
    outs, scores, enc, steps = _generate(m, torch.from_numpy(ids), torch.from_numpy(mask), cfg, R=10)
    save("g5_generate_base", input_ids=ids, attention_mask=mask, decoded=outs, scores=np.array(scores, np.float64),
         pooled=enc[::10, 0], step_scores=torch.stack([s for s, _ in steps]),
         step_tokens=torch.stack([t for _, t in steps]), num_beams=10, length_penalty=0.8, seed=1234)


def g_beam_table():
    """G5c: the reference `_generate_beam_search` driven by a synthetic logit table so that EOS, early
    `done`, eviction and the 2R>valid 'garbage candidate' paths are exercised (random weights almost
    never emit EOS).  logits(prefix) = T[pos, last_token, :] + positional mask."""
    from transformers.generation_utils import GenerationMixin
    from transformers.modeling_outputs import Seq2SeqLMOutput, BaseModelOutput
    from transformers.configuration_t5 import T5Config

    class Fake(torch.nn.Module, GenerationMixin):
        def __init__(self, table, V, Vd):
            super().__init__()
            self.table, self.V, self.Vd = table, V, Vd
            self.config = T5Config(is_encoder_decoder=True, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                                   vocab_size=Vd)
            self.p = torch.nn.Parameter(torch.zeros(1))

        def get_encoder(self):
            return lambda ids, attention_mask=None, return_dict=True: BaseModelOutput(
                last_hidden_state=torch.zeros(ids.shape[0], ids.shape[1], 4))

        def prepare_inputs_for_generation(self, input_ids, past, attention_mask, use_cache, encoder_outputs, **kw):
            return {"decoder_input_ids": input_ids, "encoder_outputs": encoder_outputs}

        def adjust_logits_during_generation(self, logits, **kwargs):
            return logits

        def get_output_embeddings(self):
            return self.p

        def forward(self, decoder_input_ids=None, encoder_outputs=None, return_dict=True):
            R, t = decoder_input_ids.shape
            b = (encoder_outputs.last_hidden_state[:, 0, 0]).long()       # query id smuggled through the encoder
            lg = self.table[b, t - 1, decoder_input_ids[:, -1]]           # [R, Vd]
            mask = torch.full((self.Vd,), -1e9)
            mask[(t - 1) * self.V + 2:(t - 1) * self.V + self.V + 2] = 0
            mask[1] = 0
            return Seq2SeqLMOutput(logits=(lg + mask)[:, None, :].expand(R, t, self.Vd))

    cases = {}
    for name, (V, maxlen, R, B, eos_boost, seed) in {
        "a": (6, 6, 4, 5, 2.5, 21),        # EOS competitive -> hyps finish at varied depths, some `done`
        "b": (6, 6, 10, 3, 1.0, 22),       # 2R=20 > 7 valid columns -> -1e9 candidates enter at step 1
        "c": (30, 10, 10, 4, 3.0, 23),     # NQ-shaped vocab (302), beam 10
    }.items():
        Vd = V * maxlen + 2
        table = synth.make_logit_table(B, maxlen, Vd, eos_boost, seed)     # regenerated from the seed by the tests
        fake = Fake(torch.from_numpy(table), V, Vd)
        ids = torch.arange(B).view(B, 1).repeat(1, 3)

        # encoder "hidden" carries the query id in [:,0,0]
        fake.get_encoder = lambda: (lambda i, attention_mask=None, return_dict=True: BaseModelOutput(
            last_hidden_state=i[:, :, None].float().repeat(1, 1, 4)))
        with torch.no_grad():
            (outs, scores), _ = quiet(fake.generate, ids, attention_mask=torch.ones_like(ids), use_cache=False,
                                      max_length=maxlen, num_beams=R, length_penalty=0.8, num_return_sequences=R,
                                      early_stopping=False, decode_embedding=2, decode_vocab_size=Vd,
                                      output_scores=True, output_encoder_embedding=True)
        cases[f"{name}_decoded"] = outs.numpy()
        cases[f"{name}_scores"] = np.array(scores, np.float64)
        cases[f"{name}_meta"] = np.array([V, maxlen, R, B, seed])
        cases[f"{name}_eos_boost"] = np.array(eos_boost)
    save("g5_beam_table", **cases)


def g_codec(main_models, main_utils):
    """G6: docid codec known answers (main_models.py:297-346, main_utils.py:70-76)."""
    args = types.SimpleNamespace(kary=30, position=1, output_vocab_size=30)
    strs = ["3-17-5", "0-0-0", "29-29-29-29", "7", "12-0-29-1-4-8-15-16-23"]
    enc = [main_models.encode_single_newid(args, s) for s in strs]
    width = max(len(e) for e in enc) + 1
    seqs = np.zeros((len(enc) + 2, width), dtype=np.int64)
    for i, e in enumerate(enc):
        seqs[i, 1:1 + len(e)] = e
    # two rows without EOS (reference prints and decodes the row whole, START included)
    seqs[-2, :] = np.arange(width) * 30 + 2 + 3
    seqs[-1, :] = np.arange(width) * 30 + 2
    dec = quiet(main_models.decode_token, args, seqs)
    args10 = types.SimpleNamespace(kary=0, position=1, output_vocab_size=10)
    enc10 = main_models.encode_single_newid(args10, "40917")
    d2 = main_utils.dec_2d(list(range(10)), 4)
    save("g6_codec", strs=np.array(strs), enc_flat=np.concatenate([np.array(e) for e in enc]),
         enc_len=np.array([len(e) for e in enc]), seqs=seqs, dec=np.array(dec), enc10=np.array(enc10),
         dec2d_flat=np.array([x for r in d2 for x in r]), dec2d_len=np.array([len(r) for r in d2]))


def g_metrics(main_metrics):
    """G7: recall()/MRR100() on a 20-line res1 TSV (main_metrics.py:194-267)."""
    g = np.random.Generator(np.random.PCG64(77))
    rows = []
    for q in range(20):
        preds = [str(x) for x in g.permutation(500)[:100]]
        if q % 3 == 0:
            gt = preds[int(g.integers(0, 100))]
        elif q % 3 == 1:
            gt = preds[0]
        else:
            gt = "9999"
        rows.append((f"query {q}", ",".join(preds), gt, "1"))
    tmp = tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False)
    for r in rows:
        tmp.write("\t".join(r) + "\n")
    tmp.close()
    rec = {}
    for k in [1, 5, 10, 20, 50, 100]:
        args = types.SimpleNamespace(res1_save_path=tmp.name, trivia=0, recall_num=[k])
        rec[k] = quiet(main_metrics.recall, args)
    mrr = quiet(main_metrics.MRR100, types.SimpleNamespace(res1_save_path=tmp.name))
    os.unlink(tmp.name)
    save("g7_metrics", rows=np.array(rows), recall_k=np.array(sorted(rec)), recall_v=np.array([rec[k] for k in sorted(rec)]),
         mrr100=np.array(mrr))


def _epoch_end_outputs(seed, n_batches, eval_batch_size, R, n_alpha, multi_gt):
    """Harness-built `outputs` of validation_step_i (main_models.py:1640-1641): cluster rows + per-alpha doc rows."""
    g = np.random.Generator(np.random.PCG64(seed))
    outputs = []
    for bi in range(n_batches):
        res, idx = [], []
        for b in range(eval_batch_size):
            text = f"query {bi}-{b}"
            clusters = [f"{int(a)}-{int(c)}" for a, c in zip(g.integers(0, 6, R), g.integers(0, 6, R))]
            sel = int(g.integers(0, 3))
            gt = clusters[int(g.integers(0, R))] if sel == 0 else (clusters[0] if sel == 1 else "9-9")
            if multi_gt and b % 2:
                gt = gt + "," + clusters[int(g.integers(0, R))]
            res.append([text, ",".join(clusters), gt, 1])
            per_alpha = []
            for a in range(n_alpha):
                docs = [str(int(x)) for x in g.permutation(60)[:R]]
                gtd = docs[int(g.integers(0, R))] if (bi + b + a) % 3 else "999"
                if multi_gt and b % 2:
                    gtd = gtd + "," + docs[int(g.integers(0, R))]
                per_alpha.append([[text, ",".join(docs), gtd]])
            idx.append(per_alpha)
        outputs.append({"inf_result_batch": res, "inf_result_batch_prob": [float(x) for x in g.standard_normal(eval_batch_size * R)],
                        "inf_index_batch": idx})
    return outputs


def g_epoch_metrics(main_models):
    """G12: T5FineTuner.validation_epoch_end (main_models.py:1643-1908) — cal_recall / cal_accuracy / cal_MRR / cal_MAP per
    alpha — run from the reference on harness-built step outputs; everything it passes to self.log is recorded."""
    import json
    cases = {}
    for name, (ite, multi) in {"two_stage": (1, True), "cluster_only": (0, True), "single_gt": (1, False)}.items():
        score_rate = [0, 0.5, 1.5]
        outputs = _epoch_end_outputs(seed=31 + len(name), n_batches=5, eval_batch_size=2, R=10,
                                     n_alpha=len(score_rate), multi_gt=multi)
        obj = main_models.T5FineTuner.__new__(main_models.T5FineTuner)
        torch.nn.Module.__init__(obj)
        logged = {}
        obj.log = lambda k, v, **kw: logged.__setitem__(k, float(v))
        obj.epoch = 0
        obj.l1_query_train_dataset = types.SimpleNamespace(epoch=0)
        obj.args = types.SimpleNamespace(begin_val_epoch=0, multiple_decoder=0, eval_batch_size=2, score_rate=score_rate,
                                         is_train_encoder=ite, train_encoder_epoch=51)
        quiet(obj.validation_epoch_end, json.loads(json.dumps(outputs)))
        cases[name + "_outputs"] = json.dumps(outputs)
        cases[name + "_args"] = json.dumps({"eval_batch_size": 2, "score_rate": score_rate, "is_train_encoder": ite})
        cases[name + "_keys"] = np.array(sorted(logged))
        cases[name + "_vals"] = np.array([logged[k] for k in sorted(logged)], np.float64)
    save("g12_epoch_metrics", **cases)


def g_rerank(main_models):
    """G4: the in-cluster rerank block of T5FineTuner.validation_step_i (main_models.py:1434-1462,1574-1637),
    executed from the reference source on harness-built attributes (object made with __new__;
    `.cuda()` patched to identity because this container has no GPU)."""
    cfg = GDRConfig.tiny()
    B, R, d = 4, 5, 64
    V = cfg.output_vocab_size
    N = 240
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=7, V=V)
    D = synth.make_corpus(N, d, cluster_size=7, seed=31)
    Q, gold = synth.make_queries(D, B, seed=32)
    g = np.random.Generator(np.random.PCG64(33))
    # decoded cluster ids per query: gold cluster first, then distinct random clusters
    dec_ids = np.zeros((B * R, depth + 2), dtype=np.int64)
    chosen = []
    for b in range(B):
        cl = [int(gold[b]) // 7] + [int(x) for x in g.permutation(len(names))[:R - 1]]
        cl = list(dict.fromkeys(cl))[:R]
        while len(cl) < R:
            cl.append(int(g.integers(0, len(names))))
        chosen.append(cl)
        for j, c in enumerate(cl):
            toks = main_models.encode_single_newid(types.SimpleNamespace(kary=V, position=1), names[c])
            dec_ids[b * R + j, 1:1 + len(toks)] = toks
    beam_scores = np.sort(g.standard_normal((B, R)).astype(np.float32) - 2.0, axis=1)[:, ::-1].copy()
    enc_hidden = np.zeros((B * R, 3, d), dtype=np.float32)
    enc_hidden[:, 0, :] = np.repeat(Q, R, axis=0)
    enc_hidden[:, 1:, :] = g.standard_normal((B * R, 2, d)).astype(np.float32)

    alphas = [0, 0.5, 1, 1.5, 2, 2.5, 3]
    args = types.SimpleNamespace(
        decode_embedding=2, position=1, max_output_length=cfg.max_output_length, hierarchic_decode=0,
        output_vocab_size=V, softmax=0, gen_method="greedy", is_train_encoder=1, multiple_decoder=0,
        num_return_sequences=R, length_penalty=0.8, label_length_cutoff=0, use_query_embed_encoder=1,
        use_query_embed_decoder_avg=0, use_query_embed_decoder_special=0, loss_func="tanh", score_rate=alphas,
        eval_batch_size=B, train_encoder_epoch=51, kary=V)

    class FakeT5:
        config = types.SimpleNamespace(hidden_size=d)

        def generate(self, *a, **k):
            return (torch.from_numpy(dec_ids), [float(x) for x in beam_scores.reshape(-1)]), \
                types.SimpleNamespace(last_hidden_state=torch.from_numpy(enc_hidden))

    ft = main_models.T5FineTuner.__new__(main_models.T5FineTuner)
    torch.nn.Module.__init__(ft)
    ft.args = args
    ft.model = FakeT5()
    ft.root, ft.cluster, ft.epoch = None, set(names), 0
    ft.id_mapping = {names[c]: [int(x) for x in members[offsets[c]:offsets[c + 1]]] for c in range(len(names))}
    ft.doc_embed = [torch.from_numpy(D[i:i + 1]) for i in range(N)]
    enc_model = main_models.EncoderModel.__new__(main_models.EncoderModel)
    torch.nn.Module.__init__(enc_model)
    enc_model.output = None
    ft.encoder = enc_model
    ft.softmax = torch.nn.Softmax(dim=-1)
    ft.tokenizer = types.SimpleNamespace(decode=lambda ids: " ".join(str(int(x)) for x in ids))
    ft.model_config = types.SimpleNamespace(pad_token_id=0, eos_token_id=1)
    batch = {"source_ids": torch.arange(B * 3).view(B, 3), "source_mask": torch.ones(B, 3, dtype=torch.long),
             "target_mask": torch.ones(B, 4, dtype=torch.long),
             "rank": [[[f"gt{b}" for b in range(B)], torch.ones(B, dtype=torch.long)]],
             "oldid": [[str(int(x)) for x in gold]]}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            out = quiet(ft.validation_step_i, batch, -1)
    finally:
        torch.Tensor.cuda = orig_cuda
    pred = np.zeros((B, len(alphas), R), dtype=np.int64)
    for b in range(B):
        for a in range(len(alphas)):
            pred[b, a] = [int(x) for x in out["inf_index_batch"][b][a][0][1].split(",")]
    cluster_strs = np.array([",".join(x[1].split(",")) for x in out["inf_result_batch"]])
    save("g4_rerank", D=D, Q=Q, gold=gold, dec_ids=dec_ids, beam_scores=beam_scores, alphas=np.array(alphas, np.float32),
         chosen=np.array(chosen), offsets=offsets, members=members, names=np.array(names), pred=pred,
         cluster_strs=cluster_strs, cluster_size=7)


def g_doc_tower():
    """G10: the doc tower `DPRContextEncoder(DPRConfig(...))` (modeling_dpr.py:146-191 over modeling_bert.py), tiny shape
    with ragged padding, and bert-base shape B=2, L=128 (pooled + a few rows)."""
    from transformers.configuration_dpr import DPRConfig
    from transformers.modeling_dpr import DPRContextEncoder
    out = {}
    for name, tiny, B, L in (("tiny", True, 5, 37), ("base", False, 2, 128)):
        bc = synth.bert_config(tiny)
        cfg = DPRConfig(vocab_size=bc["vocab_size"], hidden_size=bc["hidden_size"], num_hidden_layers=bc["num_layers"],
                        num_attention_heads=bc["num_heads"], intermediate_size=bc["d_ff"],
                        max_position_embeddings=bc["max_pos"], type_vocab_size=bc["type_vocab"], projection_dim=0)
        m = DPRContextEncoder(cfg)
        sd = synth.make_bert_state_dict(bc, seed=4321)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(("pooler" in k or "position_ids" in k) for k in missing), (missing, unexpected)
        m.eval()
        ids, mask = synth.make_tokens(B, L=L, vocab_hi=bc["vocab_size"], seed=17, min_len=3)
        with torch.no_grad():
            o = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), return_dict=True,
                  output_hidden_states=True)
            last = o.hidden_states[-1]
        out[name + "_ids"], out[name + "_mask"] = ids, mask
        out[name + "_pooled"] = o.pooler_output
        if tiny:
            out[name + "_hidden"] = last
        else:
            out[name + "_rows"] = last[:, [1, 64, 127]]
    save("g10_doc_tower", seed=4321, **out)


def g_beam_trie():
    """G11: trie-constrained beam search — the reference's earlier `generation_utils_previous.GenerationMixin`
    (un-imported by the shipped code; :714-729 hold the active constraint) with a trie built by the reference's own
    TreeBuilder (main_models.py:112-151), driven by a logit table."""
    import main_models
    from transformers.generation_utils_previous import GenerationMixin as PrevMixin
    from transformers.modeling_outputs import Seq2SeqLMOutput, BaseModelOutput
    from transformers.configuration_t5 import T5Config

    class Fake(torch.nn.Module, PrevMixin):
        def __init__(self, table, V, Vd):
            super().__init__()
            self.table, self.V, self.Vd = table, V, Vd
            self.config = T5Config(is_encoder_decoder=True, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                                   vocab_size=Vd)
            self.p = torch.nn.Parameter(torch.zeros(1))

        def get_encoder(self):
            return lambda i, attention_mask=None, return_dict=True: BaseModelOutput(
                last_hidden_state=i[:, :, None].float().repeat(1, 1, 4))

        def get_output_embeddings(self):
            return self.p

        def prepare_inputs_for_generation(self, input_ids, past, attention_mask, use_cache, encoder_outputs, **kw):
            return {"decoder_input_ids": input_ids, "encoder_outputs": encoder_outputs}

        def adjust_logits_during_generation(self, logits, **kwargs):
            return logits

        def forward(self, decoder_input_ids=None, encoder_outputs=None, return_dict=True, **_kw):
            R, t = decoder_input_ids.shape
            b = (encoder_outputs.last_hidden_state[:, 0, 0]).long()
            lg = self.table[b, t - 1, decoder_input_ids[:, -1]]
            mask = torch.full((self.Vd,), -1e9)
            mask[(t - 1) * self.V + 2:(t - 1) * self.V + self.V + 2] = 0
            mask[1] = 0
            return Seq2SeqLMOutput(logits=(lg + mask)[:, None, :].expand(R, t, self.Vd))

    cases = {}
    for name, (V, maxlen, R, B, n_ids, seed) in {"a": (6, 6, 4, 4, 60, 31), "b": (6, 6, 10, 3, 25, 32),
                                                 "c": (30, 10, 10, 3, 4000, 33)}.items():
        Vd = V * maxlen + 2
        g = np.random.Generator(np.random.PCG64(seed))
        depth_hi = maxlen - 2
        ids = set()
        while len(ids) < n_ids:
            depth = int(g.integers(2, depth_hi + 1))
            ids.add("-".join(str(int(x)) for x in g.integers(0, V, size=depth)))
        ids = sorted(ids)
        args = types.SimpleNamespace(kary=V, position=1)
        tb = main_models.TreeBuilder()
        seqs = np.zeros((len(ids), maxlen), dtype=np.int64)
        for i, s_ in enumerate(ids):
            toks = main_models.encode_single_newid(args, s_)
            seqs[i, :len(toks)] = toks
            tb.add(seqs[i].tolist(), i)
        table = synth.make_logit_table(B, maxlen, Vd, 1.5, seed)
        fake = Fake(torch.from_numpy(table), V, Vd)
        qids = torch.arange(B).view(B, 1).repeat(1, 3)
        with torch.no_grad():
            (outs, scores), _ = quiet(fake.generate, qids, attention_mask=torch.ones_like(qids), use_cache=False,
                                      max_length=maxlen, num_beams=R, length_penalty=0.8, num_return_sequences=R,
                                      early_stopping=False, decode_embedding=2, decode_vocab_size=Vd,
                                      decode_tree=tb.build(), output_scores=True, output_encoder_embedding=True)
        cases[f"{name}_seqs"] = seqs
        cases[f"{name}_decoded"] = outs.numpy()
        cases[f"{name}_scores"] = np.array(scores, np.float64)
        cases[f"{name}_meta"] = np.array([V, maxlen, R, B, seed])
    save("g11_beam_trie", **cases)


def g_cli():
    """G9: the reference argparse namespace (main.py:260-448) for no flags and for infer.sh's flags (without
    --trivia, which the reference parser rejects).  main.py itself cannot be imported (nltk / pytorch_lightning),
    so its `parsers_parser` FunctionDef is compiled out of the file with ast and executed as is."""
    import ast
    import json
    src = open(REF + "/main.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "parsers_parser"][0]
    ns = {"argparse": argparse}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF + "/main.py", "exec"), ns)
    infer = ("--decode_embedding 2 --n_gpu 1 --mode eval --query_type gtq_doc_aug_qg --adaptor_layer_num 4 "
             "--infer_ckpt CKPT --num_return_sequences 100 --tree 1 --model_info base --train_batch_size 64 "
             "--eval_batch_size 1 --test1000 0 --dropout_rate 0.1 --Rdrop 0.1 --adaptor_decode 1 --adaptor_efficient 1 "
             "--aug_query 1 --aug_query_type corrupted_query --input_dropout 1 --id_class bert_k30_c30_1 --kary 30 "
             "--output_vocab_size 30 --doc_length 64 --denoising 0 --max_output_length 10 --nq 1")
    out = {}
    for name, argv in (("default", []), ("infer_sh", infer.split())):
        old = sys.argv
        sys.argv = ["main.py"] + argv
        try:
            out[name] = vars(quiet(ns["parsers_parser"]))
        finally:
            sys.argv = old
    save("g9_cli", default=json.dumps(out["default"], sort_keys=True), infer_sh=json.dumps(out["infer_sh"], sort_keys=True),
         infer_argv=infer)


FIXTURES = ["buckets", "encoder_tiny", "encoder_base", "sim_topk", "decode_logits_tiny", "generate_tiny",
            "generate_base", "beam_table", "codec", "metrics", "rerank", "cli", "doc_tower", "beam_trie", "epoch_metrics", "dense_model"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    pkg, main_models, main_utils, main_metrics = quiet(import_reference)
    todo = a.only or FIXTURES
    for name in todo:
        print("==", name)
        if name == "codec":
            g_codec(main_models, main_utils)
        elif name == "metrics":
            g_metrics(main_metrics)
        elif name == "rerank":
            g_rerank(main_models)
        elif name == "epoch_metrics":
            g_epoch_metrics(main_models)
        else:
            globals()["g_" + name]()


if __name__ == "__main__":
    main()
