"""Pins the oracle (CPU restatement) against outputs of the reference itself (tests/golden/*.npz,
made by tests/golden/make_golden.py from the imported reference).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden, order_insensitive_topk_match
from gdr_amd.config import GDRConfig
from gdr_amd import synth
from oracle import t5_ref, beam_ref, retrieval_ref, codec_ref

torch.set_grad_enabled(False)


def test_relative_position_buckets_bit_exact():
    g = golden("g2_buckets")
    rel = torch.arange(128)[None, :] - torch.arange(128)[:, None]
    assert np.array_equal(t5_ref.relative_position_bucket(rel, True, 32).numpy(), g["bidirectional"].astype(np.int64))
    assert np.array_equal(t5_ref.relative_position_bucket(rel, False, 32).numpy(), g["unidirectional"].astype(np.int64))


def test_encoder_tiny_matches_reference():
    g = golden("g1_encoder_tiny")
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    h, pb = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]),
                                   return_bias=True)
    np.testing.assert_allclose(pb.numpy(), g["position_bias"], rtol=0, atol=0)
    np.testing.assert_allclose(h.numpy(), g["last_hidden_state"], rtol=1e-5, atol=1e-5)


@pytest.mark.timeout(600)
def test_encoder_base_matches_reference():
    g = golden("g1_encoder_base")
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]), with_decoder=False)
    h = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    np.testing.assert_allclose(h[:, 0].numpy(), g["pooled"], rtol=2e-4, atol=2e-4)
    rc = g["sample_rc"]
    np.testing.assert_allclose(h[rc[:, 0], rc[:, 1]].numpy(), g["sample_rows"], rtol=2e-4, atol=2e-4)


def test_sim_topk_c1_matches_reference():
    g = golden("g3_sim_topk")
    D = synth.make_corpus(1000, 768)
    Q, gold = synth.make_queries(D, 128)
    assert np.array_equal(gold, g["gold"])
    v, i = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), 10)
    order_insensitive_topk_match(g["values"], g["indices"], v.numpy(), i.numpy(), 1e-5)
    # blocked variant (used by the CPU baseline) is the same arithmetic
    v2, i2 = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), 10, block=32)
    order_insensitive_topk_match(g["values"], g["indices"], v2.numpy(), i2.numpy(), 1e-5)


def test_pooler_contract():
    g = golden("g3_sim_topk")
    out = retrieval_ref.cls_pool(torch.from_numpy(g["pool_hidden"]), torch.from_numpy(g["pool_w"]),
                                 torch.from_numpy(g["pool_b"]), normalize=True)
    np.testing.assert_allclose(out.numpy(), g["pool_out"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["pool", "cls"])
def test_dense_model_matches_reference(mode):
    """dense.DenseModel(lm_q, lm_p, pooler).eval()(query=, passage=) run from the reference (g13): the oracle's encoder +
    cls_pool + compute_similarity + topk reproduce q_reps / p_reps / scores / top-3."""
    g = golden("g13_dense_model")
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    hq = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(g["q_ids"]), torch.from_numpy(g["q_mask"]))
    hp = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(g["p_ids"]), torch.from_numpy(g["p_mask"]))
    T = torch.from_numpy
    if mode == "pool":
        q = retrieval_ref.cls_pool(hq, T(g["wq"]), T(g["bq"]), normalize=True)
        p = retrieval_ref.cls_pool(hp, T(g["wp"]), T(g["bp"]), normalize=True)
    else:
        q, p = retrieval_ref.cls_pool(hq), retrieval_ref.cls_pool(hp)
    torch.testing.assert_close(q, T(g[mode + "_q_reps"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(p, T(g[mode + "_p_reps"]), rtol=1e-4, atol=1e-5)
    sc = retrieval_ref.compute_similarity(q, p)
    torch.testing.assert_close(sc, T(g[mode + "_scores"]), rtol=1e-4, atol=1e-5)
    v, i = sc.topk(3, dim=1)
    assert np.array_equal(i.numpy(), g[mode + "_top_i"])


def test_decode_logits_tiny_matches_reference():
    g = golden("g8_decode_logits_tiny")
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    enc = t5_ref.encoder_forward(sd, cfg, ids, mask)
    np.testing.assert_allclose(enc.numpy(), g["enc"], rtol=1e-5, atol=1e-5)
    dec = torch.from_numpy(g["decoder_input_ids"])
    h = t5_ref.decoder_forward(sd, cfg, dec, enc, mask)
    a = t5_ref.adaptor_forward(sd, cfg, dec)
    full = t5_ref.head_full(sd, cfg, h, a)
    ref = g["logits"]
    np.testing.assert_allclose(full.numpy(), ref, rtol=2e-5, atol=2e-5)
    # the restricted head (last position, valid columns) gives the same row
    t = dec.shape[1]
    last = t5_ref.head_last_restricted(sd, cfg, h[:, -1], a[:, -1], t - 1)
    np.testing.assert_allclose(last.numpy(), ref[:, -1], rtol=2e-5, atol=2e-5)
    masked = np.ones(cfg.decode_vocab_size, bool)
    masked[t5_ref.valid_columns(t - 1, cfg.output_vocab_size)] = False
    assert (last.numpy()[:, masked] == np.float32(-1e9)).all() and (ref[:, -1][:, masked] == np.float32(-1e9)).all()


def _check_generate(g, cfg, sd, restricted):
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    R = int(g["num_beams"])
    trace = []
    (dec, scores), enc = beam_ref.generate(sd, cfg, ids, mask, R, length_penalty=float(g["length_penalty"]),
                                           restricted_head=restricted, trace=trace)
    assert np.array_equal(dec.numpy(), g["decoded"])
    np.testing.assert_allclose(np.array(scores), g["scores"], rtol=1e-5, atol=1e-5)
    st = torch.stack([s for s, _ in trace]).numpy()
    ref = g["step_scores"]
    finite = ref > -1e8
    np.testing.assert_allclose(st[finite], ref[finite], rtol=2e-5, atol=2e-5)
    tk = torch.stack([t for _, t in trace]).numpy()
    assert np.array_equal(tk[finite], g["step_tokens"][finite])
    return enc


def test_generate_tiny_matches_reference():
    g = golden("g5_generate_tiny")
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    enc = _check_generate(g, cfg, sd, restricted=False)
    np.testing.assert_allclose(enc.numpy(), g["enc"], rtol=1e-5, atol=1e-5)
    _check_generate(g, cfg, sd, restricted=True)


@pytest.mark.timeout(900)
def test_generate_base_matches_reference():
    g = golden("g5_generate_base")
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    enc = _check_generate(g, cfg, sd, restricted=True)
    np.testing.assert_allclose(enc[::int(g["num_beams"]), 0].numpy(), g["pooled"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_beam_search_table_matches_reference(case):
    g = golden("g5_beam_table")
    V, maxlen, R, B, seed = [int(x) for x in g[f"{case}_meta"]]
    Vd = V * maxlen + 2
    table = torch.from_numpy(synth.make_logit_table(B, maxlen, Vd, float(g[f"{case}_eos_boost"]), seed))
    qid = torch.arange(B).repeat_interleave(R)

    def step(seq):
        t = seq.shape[1]
        lg = table[qid, t - 1, seq[:, -1]]
        return lg + t5_ref.positional_mask(t, Vd, V)[t - 1]

    dec, scores = beam_ref.beam_search(step, B, R, Vd, maxlen, 0.8)
    ref_dec, ref_sc = g[f"{case}_decoded"], g[f"{case}_scores"]
    finite = ref_sc > -1e8
    assert finite.all(), "fixture should end with finite hypotheses only"
    np.testing.assert_allclose(np.array(scores), ref_sc, rtol=1e-6, atol=1e-6)
    assert np.array_equal(dec.numpy(), ref_dec)


def test_codec_known_answers():
    g = golden("g6_codec")
    off = 0
    for s, n in zip(g["strs"], g["enc_len"]):
        assert codec_ref.encode_single_newid(str(s)) == g["enc_flat"][off:off + n].tolist()
        off += n
    assert codec_ref.decode_token(g["seqs"]) == [str(x) for x in g["dec"]]
    assert codec_ref.encode_single_newid("40917", kary=0) == g["enc10"].tolist()
    d2 = codec_ref.dec_2d(list(range(10)), 4)
    assert [len(r) for r in d2] == g["dec2d_len"].tolist()
    assert [x for r in d2 for x in r] == g["dec2d_flat"].tolist()


def test_metrics_known_answers():
    g = golden("g7_metrics")
    rows = [tuple(str(x) for x in r) for r in g["rows"]]
    rec = codec_ref.recall_from_rows(rows, g["recall_k"].tolist())
    for k, v in zip(g["recall_k"], g["recall_v"]):
        assert rec[int(k)] == pytest.approx(float(v), abs=0)
    assert codec_ref.mrr100_from_rows(rows) == pytest.approx(float(g["mrr100"]), abs=1e-15)


def test_rerank_matches_reference():
    g = golden("g4_rerank")
    B, R = g["chosen"].shape
    names = [str(x) for x in g["names"]]
    offsets, members = g["offsets"], g["members"]
    # decode the cluster ids exactly as validation_step_i does (decode_token + dec_2d)
    dec = codec_ref.dec_2d(codec_ref.decode_token(g["dec_ids"], output_vocab_size=6, kary=6), R)
    assert [",".join(d) for d in dec] == [str(x) for x in g["cluster_strs"]]
    name_to_c = {n: i for i, n in enumerate(names)}
    mem_q, num_q = [], []
    for b in range(B):
        mem, num = [], []
        for s in dec[b]:
            c = name_to_c[s]
            seg = members[offsets[c]:offsets[c + 1]].tolist()
            mem += seg
            num.append(len(seg))
        mem_q.append(mem)
        num_q.append(num)
    out = retrieval_ref.rerank(torch.from_numpy(g["Q"]), torch.from_numpy(g["D"]), mem_q, num_q,
                               g["beam_scores"].tolist(), g["alphas"].tolist(), R)
    for b in range(B):
        for a in range(len(g["alphas"])):
            assert out[b][a][1].tolist() == g["pred"][b, a].tolist()


def test_doc_tower_matches_reference():
    from oracle import bert_ref
    g = golden("g10_doc_tower")
    for name, tiny in (("tiny", True), ("base", False)):
        bc = synth.bert_config(tiny)
        sd = synth.make_bert_state_dict(bc, seed=int(g["seed"]))
        h, pooled = bert_ref.bert_forward(sd, bc, torch.from_numpy(g[name + "_ids"]), torch.from_numpy(g[name + "_mask"]))
        np.testing.assert_allclose(pooled.numpy(), g[name + "_pooled"], rtol=2e-5, atol=2e-5)
        if tiny:
            np.testing.assert_allclose(h.numpy(), g["tiny_hidden"], rtol=2e-5, atol=2e-5)
        else:
            np.testing.assert_allclose(h[:, [1, 64, 127]].numpy(), g["base_rows"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_trie_constrained_beam_matches_reference(case):
    """The trie mode of generation_utils_previous.py:714-729 (restated in beam_ref.trie_mask)."""
    g = golden("g11_beam_trie")
    V, maxlen, R, B, seed = [int(x) for x in g[f"{case}_meta"]]
    Vd = V * maxlen + 2
    table = torch.from_numpy(synth.make_logit_table(B, maxlen, Vd, 1.5, seed))
    qid = torch.arange(B).repeat_interleave(R)
    tree = beam_ref.build_trie(g[f"{case}_seqs"].tolist())

    def step(seq):
        t = seq.shape[1]
        return table[qid, t - 1, seq[:, -1]] + t5_ref.positional_mask(t, Vd, V)[t - 1]

    dec, scores = beam_ref.beam_search(step, B, R, Vd, maxlen, 0.8, decode_tree=tree)
    np.testing.assert_allclose(np.array(scores), g[f"{case}_scores"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(dec.numpy(), g[f"{case}_decoded"])
    # every returned id is a real docid of the trie
    valid = {tuple(int(t) for t in s if t != 0) for s in g[f"{case}_seqs"].tolist()}
    for row in dec.numpy():
        toks = [int(t) for t in row[1:] if t != 0]
        assert tuple(toks) in valid
