"""GPU parity of the named surface at the reference's own entry points and at BASELINE.json's configuration shapes:

  * `python -m gdr_amd.main --mode eval` with infer.sh's flags (GDR_model/main.py:115-250, infer.sh:10-15): the res1 TSV
    (and the doc-level TSV of the two-stage path) row for row against the oracle composition
    beam_ref.generate -> codec_ref.decode_token -> retrieval_ref.rerank;
  * dense.DenseModel / DensePooler (dense.py:10-54) through encode_query -> compute_similarity -> search, against
    fixtures made by running the reference classes (g13, g3);
  * config C2 — all 512 query rows of the 320 000 x 768 top-100 against the oracle;
  * config C4 — 4096 queries x 8 shards of 40 000 rows: packed exchange form + merge == single-shard search, rows vs oracle;
  * config C3 at full size — 320 000 docs, 64 queries, beam 10, t5-base two-stage retrieval, eight queries vs the oracle;
  * config C5 composed — 1M x 768 bf16 corpus, beam 30, bf16 mode through encoder, decode and rerank, B = 64;
  * a Lightning-style checkpoint dict ({"state_dict": {"model.…", "encoder.model.…"}}, main.py:121-126) through
    GDRModel / EncoderModel.from_state_dict and through `--infer_ckpt`.
"""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import REPO, beam_cut_explains_absence, golden, hypothesis_lists_match, order_insensitive_topk_match, ranked_lists_match
from gdr_amd.config import GDRConfig
from gdr_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1e-4

# GDR_model/infer.sh:10-15 as written (BEAM_SIZE / INFER_CKPT substituted per test)
INFER_SH = ("--decode_embedding 2 --n_gpu 1 --mode eval --query_type gtq_doc_aug_qg --adaptor_layer_num 4 "
            "--tree 1 --model_info base --train_batch_size 64 --test1000 0 --dropout_rate 0.1 --Rdrop 0.1 "
            "--adaptor_decode 1 --adaptor_efficient 1 --aug_query 1 --aug_query_type corrupted_query --input_dropout 1 "
            "--id_class bert_k30_c30_1 --kary 30 --output_vocab_size 30 --doc_length 64 --denoising 0 "
            "--max_output_length 10 --trivia 0 --nq 1").split()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run_main(argv, timeout=1500):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "gdr_amd.main"] + argv, cwd=REPO, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def _read_tsv(path):
    return [line.rstrip("\n").split("\t") for line in open(path)]


@pytest.fixture(scope="module")
def base_weights():
    cfg = GDRConfig.base()
    return cfg, synth.make_state_dict(cfg, seed=1234)


@pytest.mark.parametrize("R,n_q,bs,constrain", [(10, 8, 4, 0), (10, 8, 4, 1), (100, 2, 1, 0), (100, 2, 1, 1)])
def test_main_eval_entry_point_vs_oracle_composition(tmp_path, base_weights, R, n_q, bs, constrain):
    """The entry point the reference names (main.py --mode eval with infer.sh's flags) on a reduced synthetic corpus
    (30 000 docs): stage-1 rows of the res1 TSV == oracle beam decode + decode_token, and with the trie constraint
    (valid cluster ids -> real candidates) the doc-level rows for every alpha == the oracle rerank — also at infer.sh's own
    beam width (BEAM = 100, infer.sh:10-15: 1 200 candidates per query, topk(100), main_models.py:1625).  Rows may
    permute only inside tolerance-tie groups of the oracle's scores."""
    from oracle import beam_ref, codec_ref, retrieval_ref
    cfg, sd = base_weights
    N = 30000
    res1 = str(tmp_path / "res1.tsv")
    out = _run_main(INFER_SH + ["--infer_ckpt", "", "--num_return_sequences", str(R), "--eval_batch_size", str(bs),
                                "--corpus_rows", str(N), "--n_queries", str(n_q), "--constrain_tree", str(constrain),
                                "--res1_save_path", res1])
    assert "recall@1:" in out and "MRR100:" in out
    # ---- oracle composition on the same seeded inputs
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
    ids, mask = synth.make_tokens(n_q, L=40, seed=11)
    tree = beam_ref.build_trie([codec_ref.encode_single_newid(s, kary=30) for s in names]) if constrain else None
    (rd, rs), enc_x = beam_ref.generate(sd, cfg, torch.from_numpy(ids), torch.from_numpy(mask), R, max_length=10,
                                        restricted_head=True, decode_tree=tree)
    dec = codec_ref.dec_2d(codec_ref.decode_token(rd.numpy(), output_vocab_size=30, kary=30), R)
    rs2 = np.array(rs, np.float64).reshape(n_q, R)
    D = synth.make_corpus(N, cfg.d_model)
    _, gold = synth.make_queries(D, n_q)
    rows = {r[0]: r for r in _read_tsv(res1)}
    assert [r[0] for r in _read_tsv(res1)] == sorted(rows), "res1 is sorted by query like the reference's sort_values"
    assert len(rows) == n_q
    for q in range(n_q):
        r = rows[f"q{q}"]
        ranked_lists_match(dec[q], rs2[q], r[1].split(","), TOL)
        assert r[2] == names[int(gold[q]) // 12] and r[3] == "1"
    if not constrain:
        return
    # ---- stage 2: every decoded cluster is a real one, so the rerank sees R * 12 candidates per query
    look = {n: i for i, n in enumerate(names)}
    mem_q = [[m for s in row for m in members[offsets[look[s]]:offsets[look[s] + 1]].tolist()] for row in dec]
    num_q = [[12] * R for _ in dec]
    alphas = [0, 0.5, 1, 1.5, 2, 2.5, 3]
    ref = retrieval_ref.rerank(enc_x[::R][:, 0], torch.from_numpy(D), mem_q, num_q, rs2.astype(np.float32).tolist(),
                               alphas, R)
    docs = _read_tsv(res1 + ".docs.tsv")
    assert len(docs) == n_q * len(alphas)
    got = {(r[0], float(r[1])): r for r in docs}
    for q in range(n_q):
        for ai, al in enumerate(alphas):
            r = got[(f"q{q}", float(al))]
            ranked_lists_match([str(x) for x in ref[q][ai][1].tolist()], ref[q][ai][0].numpy(), r[2].split(","), TOL)
            assert r[3] == str(int(gold[q]))
    assert "stage 2 (in-cluster rerank)" in out


def test_main_eval_sharded_two_stage_under_one_rank_rccl_equals_unsharded(tmp_path):
    """The sharded two-stage path as ONE path, started by the product's own launcher (gdr_amd/launch.py — what
    `main.py --n_gpu N` and `bench.py --gpus N` call): a 1-rank RCCL group under torch.distributed.run runs
    GDRRetriever(sharded=ShardedIndex) — generate -> gdr_cluster_candidates -> rerank_own (wire pack, all-gather, per-shard
    GDR_RERANK_POSITIONS lists, all-to-all, gdr_topk_merge_packed, positions -> ids) — through the entry point, with a short
    last batch (padding rows dropped), and writes the same res1 and doc-level TSVs, byte for byte, as the plain one-process
    run (main_models.py:1434-1462,1574-1637; Data_process/NQ_dataset/bert/bert_NQ.sh:5-12 for the per-GPU launch)."""
    from gdr_amd import launch
    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    argv = INFER_SH + ["--infer_ckpt", "", "--num_return_sequences", "10", "--eval_batch_size", "4", "--corpus_rows", "30000",
                       "--n_queries", "10", "--constrain_tree", "1"]
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cwd = os.getcwd()
    os.chdir(REPO)
    try:
        rc, text = launch.spawn_ranks(1, argv + ["--res1_save_path", a], module="gdr_amd.main", env=env, relay=False, timeout=1500)
    finally:
        os.chdir(cwd)
    assert rc == 0, text[-4000:]
    assert "GDR_RESULT " in text and "1 GPU(s)" in text
    _run_main(argv + ["--res1_save_path", b])
    assert open(a).read() == open(b).read() and len(_read_tsv(a)) == 10
    assert open(a + ".docs.tsv").read() == open(b + ".docs.tsv").read() and len(_read_tsv(a + ".docs.tsv")) == 70


@pytest.mark.parametrize("two_stage", [1, 0])
def test_main_eval_n_gpu_2_ranks_sharing_one_gpu_equals_the_one_process_run(tmp_path, two_stage):
    """`--n_gpu 2` with REAL compute on both ranks: the launcher starts two rank processes that share this box's one GPU
    (`--dist_backend gloo`: RCCL refuses two ranks on one device; gloo stages the same collectives through the host).  Rank r holds
    rows [lo, hi) of the corpus (cluster-aligned), decodes spans r, r + 2, ... (10 queries in batches of 4: three spans — rank 1 runs
    out and repeats its last one, the short last batch is padded), `rerank_own` exchanges the wire rows (world 2: every query's
    candidates really are split over two shards), merges, maps positions back; rank 0 gathers and writes.  res1 and doc-level
    TSVs byte-identical to the one-process run — the N > 1 logic end to end on the HIP kernels (main_models.py:1434-1462,
    1574-1637; per-GPU launch: bert_NQ.sh:5-12)."""
    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    argv = INFER_SH + ["--infer_ckpt", "", "--num_return_sequences", "10", "--eval_batch_size", "4", "--corpus_rows", "30000",
                       "--n_queries", "10", "--constrain_tree", "1", "--is_train_encoder", str(two_stage)]
    argv2 = [x for x in argv]
    argv2[argv2.index("--n_gpu") + 1] = "2"
    out = _run_main(argv2 + ["--dist_backend", "gloo", "--res1_save_path", a])
    assert "2 GPU(s)" in out and "GDR_RESULT " in out
    _run_main(argv + ["--res1_save_path", b])
    assert open(a).read() == open(b).read() and len(_read_tsv(a)) == 10
    if two_stage:
        assert open(a + ".docs.tsv").read() == open(b + ".docs.tsv").read() and len(_read_tsv(a + ".docs.tsv")) == 70
    else:            # --is_train_encoder 0: stage 1 only (main.py:171-238) — data-parallel generate(), rows reassembled in span order
        assert not os.path.exists(a + ".docs.tsv")


def test_main_eval_missing_checkpoint_is_an_error(tmp_path):
    """A non-empty --infer_ckpt that does not exist must fail (the reference's torch.load raises, main.py:121), not fall
    back to random weights."""
    env = dict(os.environ, PYTHONPATH=REPO)
    p = subprocess.run([sys.executable, "-m", "gdr_amd.main"] + INFER_SH + ["--infer_ckpt", "ckpt file", "--n_queries", "1"],
                       cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode != 0 and "does not exist" in p.stdout


def test_dense_model_and_pooler_vs_reference_golden(dev):
    """dense.DenseModel(lm_q, lm_p, pooler) of the reference (g13: run from dense.py / encoder.py over the reference T5
    encoder) vs the product classes on the GPU: encode_query / encode_passage / forward().scores / search()."""
    from gdr_amd.modeling import DenseModel, DensePooler, GDRModel
    g = golden("g13_dense_model")
    cfg = GDRConfig.tiny()
    lm = GDRModel(cfg, synth.make_state_dict(cfg, seed=int(g["seed"])), dev, with_decoder=False).get_encoder()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    qry = {"input_ids": T(g["q_ids"]), "attention_mask": T(g["q_mask"])}
    psg = {"input_ids": T(g["p_ids"]), "attention_mask": T(g["p_mask"])}
    pool = DensePooler(T(g["wq"]), T(g["bq"]), T(g["wp"]), T(g["bp"]), normalize=True)
    for mode, pooler in (("pool", pool), ("cls", None)):
        m = DenseModel(lm, lm, pooler=pooler)
        o = m(query=qry, passage=psg)
        assert o.loss is None
        np.testing.assert_allclose(o.q_reps.cpu().numpy(), g[mode + "_q_reps"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(o.p_reps.cpu().numpy(), g[mode + "_p_reps"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(o.scores.cpu().numpy(), g[mode + "_scores"], rtol=1e-4, atol=2e-5)
        assert torch.equal(m.encode_query(qry), o.q_reps) and torch.equal(m.encode_passage(psg), o.p_reps)
        v, i = m.search(o.q_reps, o.p_reps, 3)                            # compute_similarity + topk fused
        assert i.dtype == torch.int64
        order_insensitive_topk_match(g[mode + "_top_v"], g[mode + "_top_i"].astype(np.int64), v.cpu().numpy(),
                                     i.cpu().numpy(), TOL)
        only_q = m(query=qry)
        assert only_q.p_reps is None and only_q.scores is None and torch.equal(only_q.q_reps, o.q_reps)
    # the pooler alone on the g3 fixture (random hidden states through dense.DensePooler)
    g3 = golden("g3_sim_topk")
    p3 = DensePooler(T(g3["pool_w"]), T(g3["pool_b"]), normalize=True)
    np.testing.assert_allclose(p3(q=T(g3["pool_hidden"])).cpu().numpy(), g3["pool_out"], rtol=1e-4, atol=2e-5)
    with pytest.raises(ValueError):
        p3()


def test_c2_all_512_rows_vs_oracle(dev):
    """Config C2 at its own size: every one of the 512 query rows of the 320 000 x 768 top-100 against the oracle
    (the CPU product is a few seconds); ids may permute only inside tolerance-tie groups."""
    from gdr_amd import ops
    from oracle import retrieval_ref
    D = synth.make_corpus(320000, 768)
    Q, gold = synth.make_queries(D, 512)
    v, i, st = ops.sim_topk(torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev), 100, return_status=True)
    assert int(st.sum().item()) == 0
    rv, ri = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), 100, block=128)
    gi = i.cpu().numpy().astype(np.int64)
    permuted = order_insensitive_topk_match(rv.numpy(), ri.numpy(), v.cpu().numpy(), gi, TOL)
    identical = int((gi == ri.numpy()).all(axis=1).sum())
    assert identical >= 480 and permuted <= 64, (identical, permuted)     # near-ties are rare, not the rule
    hit = lambda idx, k: float(np.mean([gold[b] in idx[b, :k] for b in range(512)]))   # noqa: E731
    for k in (1, 10, 100):
        assert abs(hit(gi, k) - hit(ri.numpy(), k)) * 100 <= 0.1          # Recall@k within +-0.1 (north_star)


def test_c4_shape_4096_queries_8_shards_packed_merge(dev):
    """Config C4's shape on one GPU: 4096 queries against 8 row shards of 40 000 docs (cluster-aligned), each shard's
    (values, ids, status) in the packed wire form, merged — must equal the single-shard search over all 320 000 rows
    (rows are independent: same scores, same tie rule), and 8 spread rows must match the oracle."""
    from gdr_amd import ops
    from gdr_amd.dist import shard_bounds
    from oracle import retrieval_ref
    N, G, B, k = 320000, 8, 4096, 100
    D = synth.make_corpus(N, 768)
    Q, _ = synth.make_queries(D, B, seed=13)
    Dd, Qd = torch.from_numpy(D).to(dev), torch.from_numpy(Q).to(dev)
    ws = ops.Workspace(dev)
    packs = []
    for r in range(G):
        lo, hi = shard_bounds(N, G, r, cluster_size=12)
        assert 39990 <= hi - lo <= 40008 and lo % 12 == 0
        v, i, st = ops.sim_topk(Qd, Dd[lo:hi], k, idx_offset=lo, workspace=ws, return_status=True)
        assert int(i.min()) >= lo and int(i.max()) < hi
        packs.append(ops.topk_pack(v, i, st))
    mv, mi, ms = ops.topk_merge_packed(torch.stack(packs), return_status=True)
    assert int(ms.sum().item()) == 0
    fv, fi, fs = ops.sim_topk(Qd, Dd, k, workspace=ws, return_status=True)
    assert int(fs.sum().item()) == 0
    order_insensitive_topk_match(fv.cpu().numpy(), fi.cpu().numpy().astype(np.int64), mv.cpu().numpy(),
                                 mi.cpu().numpy().astype(np.int64), 1e-6)
    assert float((mi == fi).all(dim=1).float().mean()) > 0.99
    rows = np.linspace(0, B - 1, 8).astype(np.int64)
    rv, ri = retrieval_ref.sim_topk(torch.from_numpy(Q[rows]), torch.from_numpy(D), k)
    order_insensitive_topk_match(rv.numpy(), ri.numpy(), mv[rows].cpu().numpy(), mi[rows].cpu().numpy().astype(np.int64), TOL)
    # a flagged shard reaches the merged status of exactly that query
    st2 = torch.zeros(B, dtype=torch.int32, device=dev)
    st2[77] = 1
    lo3, hi3 = shard_bounds(N, G, 3, cluster_size=12)
    v3, i3 = ops.sim_topk(Qd, Dd[lo3:hi3], k, idx_offset=lo3, workspace=ws)
    packs[3] = ops.topk_pack(v3, i3, st2)
    _, mi2, ms2 = ops.topk_merge_packed(torch.stack(packs), return_status=True)
    assert ms2.nonzero().flatten().tolist() == [77] and torch.equal(mi2, mi)


def test_c3_full_size_two_stage_vs_oracle(dev, base_weights):
    """Config C3 at full size: 320 000 docs, a batch of 64 queries, beam 10, t5-base: validation_step_i (decode ->
    id_mapping -> in-cluster rerank) vs the oracle composition on eight of the queries.  Random weights decode
    full-length rows that name no cluster, so every decoded string is given a real 12-doc cluster of the corpus."""
    from gdr_amd import codec
    from gdr_amd.modeling import GDRModel, GDRRetriever
    from oracle import beam_ref, codec_ref, retrieval_ref
    cfg, sd = base_weights
    N, B, R = 320000, 64, 10
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
    Dn = synth.make_corpus(N, cfg.d_model)
    D = torch.from_numpy(Dn).to(dev)
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    batch = {"source_ids": torch.from_numpy(ids).to(dev), "source_mask": torch.from_numpy(mask).to(dev)}
    args = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8,
                                 kary=30, position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
    model = GDRModel(cfg, sd, dev)
    (outs, _), _ = model.generate(batch["source_ids"], attention_mask=batch["source_mask"], max_length=10, num_beams=R,
                                  length_penalty=0.8, num_return_sequences=R, output_scores=True)
    first = {"clusters": codec.dec_2d(codec.decode_token(args, outs.cpu().numpy()), R)}
    strs = sorted({s for row in first["clusters"] for s in row})
    assert len(strs) > B                                      # the batch really decodes many different ids
    stride = len(names) // len(strs)                          # spread the decoded ids over the whole corpus
    renamed = list(names)
    for j, s in enumerate(strs):
        renamed[j * stride] = s
    index = codec.ClusterIndex(renamed, offsets, members)
    out = GDRRetriever(model, D, index, args).validation_step_i(batch)
    assert out["clusters"] == first["clusters"]
    nq = 8
    (rd, rs), enc_x = beam_ref.generate(sd, cfg, torch.from_numpy(ids[:nq]), torch.from_numpy(mask[:nq]), R,
                                        max_length=10, restricted_head=True)
    dec = codec_ref.dec_2d(codec_ref.decode_token(rd.numpy(), output_vocab_size=30, kary=30), R)
    rs2 = np.array(rs, np.float64).reshape(nq, R)
    look = {n: i for i, n in enumerate(renamed)}
    for q in range(nq):
        ranked_lists_match(dec[q], rs2[q], out["clusters"][q], TOL)
    got_scores = np.array(out["inf_result_batch_prob"]).reshape(B, R)[:nq]
    np.testing.assert_allclose(got_scores, rs2, rtol=1e-4, atol=1e-4)
    # Stage 2 on every one of the nq queries.  Where the product's beam order equals the oracle's, the oracle rerank runs on
    # the oracle's own stage-1 output (end to end).  Where two beams swapped inside a tolerance tie (stage 1 above allows
    # exactly that), the candidates are laid out in the product's order with the product's beam scores — already held to the
    # oracle's within 1e-4 — so that stage 2 is still compared value by value and id by id, never skipped.
    end_to_end = 0
    for q in range(nq):
        same = out["clusters"][q] == dec[q]
        end_to_end += int(same)
        order = dec[q] if same else out["clusters"][q]
        bs = (rs2[q] if same else got_scores[q]).astype(np.float32)
        mem = [m for s_ in order for m in (members[offsets[look[s_]]:offsets[look[s_] + 1]].tolist() if s_ in look else [])]
        num = [12 if s_ in look else 0 for s_ in order]
        ref = retrieval_ref.rerank(enc_x[q * R:q * R + 1][:, 0], torch.from_numpy(Dn), [mem], [num], [bs.tolist()],
                                   args.score_rate, R)[0]
        for a in range(len(args.score_rate)):
            ranked_lists_match([str(x) for x in ref[a][1].tolist()], ref[a][0].numpy(), out["doc_ids"][q][a], TOL)
            np.testing.assert_allclose(out["rerank_values"][q, a].cpu().numpy(), ref[a][0].numpy(), rtol=1e-4, atol=1e-4)
    assert end_to_end >= nq - 2, end_to_end                    # swapped near-ties are the exception


def test_c5_composed_bf16_two_stage_on_1m_corpus(dev, base_weights):
    """BASELINE config C5 composed on one GPU: a 1 000 000 x 768 bf16 corpus (83 334 clusters of 12), a batch of 64 queries,
    beam 30, the bf16 precision mode end to end — encoder (bf16 linears) -> generate() (bf16, prefix table, trie-constrained
    so that every decoded id names a cluster) -> device cluster lookup -> in-cluster rerank over the bf16 corpus
    (gdr_rerank_topk_bf16; main_models.py:1574-1637).  The reference has no bf16 mode, so parity is stated per stage
    against the oracle applied to the same bf16-rounded operands:
      stage 1 (8 of the 64 queries): the oracle's emulation of the decode path's rounding points on the GPU's own encoder
              states; scores within 3e-2 of it; the score gap on shared hypotheses is MEASURED, asserted <= 5e-3 and used as
              the absolute tie tolerance of the id rule (conftest.hypothesis_lists_match): a hypothesis may sit at another
              rank than the emulation's only inside a group of hypotheses whose emulation scores chain closer than twice that
              gap, in every group including the last; an id the emulation's list lacks must be explained by a TIE at a cut of the
              emulation's own beam search (final or intermediate: conftest.beam_cut_explains_absence replays its per-step
              selection from the oracle's trace); >= 95 % of the ids are the emulation's; the group sizes are printed;
      stage 2 (all 64 queries): the oracle rerank on the bf16-rounded corpus rows and the GPU's stage-1 output: values
              to 1e-4, ids exact outside fp32 tolerance ties."""
    from gdr_amd import codec, ops
    from gdr_amd.modeling import GDRModel, GDRRetriever
    from oracle import beam_ref, codec_ref, retrieval_ref, t5_ref
    cfg, sd = base_weights
    N, B, R, V = 1000000, 64, 30, 30
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=V)
    assert depth == 4 and len(names) == 83334
    Dn = synth.make_corpus(N, cfg.d_model)
    D16 = ops.to_bf16(torch.from_numpy(Dn).to(dev))
    del Dn
    torch.cuda.empty_cache()
    ids, mask = synth.make_tokens(B, L=40, seed=23)
    batch = {"source_ids": torch.from_numpy(ids).to(dev), "source_mask": torch.from_numpy(mask).to(dev)}
    args = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=V, max_output_length=10, length_penalty=0.8,
                                 kary=V, position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
    trie = codec.Trie.from_docids(names, V)
    model = GDRModel(cfg, sd, dev, trie=trie, prefix_trie=trie, dtype=torch.bfloat16, ragged=True)
    assert model.prefix_table.n_levels == 5
    retr = GDRRetriever(model, D16, codec.ClusterIndex(names, offsets, members), args)
    state = retr._step_launch(batch)
    out = retr._step_finish(state)
    look = {n: i for i, n in enumerate(names)}
    assert all(s in look for row in out["clusters"] for s in row), "constrained beams decode real clusters only"
    got_scores = np.array(out["inf_result_batch_prob"], np.float64).reshape(B, R)
    # ---- stage 1 vs the oracle's bf16 emulation, on the GPU's own encoder states
    nq = 8
    enc_cpu = state["enc_h"][:nq].cpu()
    idx = torch.arange(nq).view(-1, 1).repeat(1, R).view(-1)
    enc_x, mask_x = enc_cpu.index_select(0, idx), torch.from_numpy(mask[:nq]).index_select(0, idx)
    tree = beam_ref.build_trie([codec_ref.encode_single_newid(s, kary=V) for s in names])

    def step(seq):
        with t5_ref.bf16_linears():
            return t5_ref.decode_logits(sd, cfg, seq, enc_x, mask_x, restricted=True)

    trace, ptrace = [], []
    rd, rs = beam_ref.beam_search(step, nq, R, cfg.decode_vocab_size, 10, 0.8, R, decode_tree=tree, trace=trace, prefix_trace=ptrace)
    dec = codec_ref.dec_2d(codec_ref.decode_token(rd.numpy(), output_vocab_size=V, kary=V), R)
    rs2 = np.array(rs, np.float64).reshape(nq, R)
    np.testing.assert_allclose(got_scores[:nq], rs2, rtol=3e-2, atol=3e-2)
    # the measured score gap on the hypotheses both sides returned IS the bf16 noise of a hypothesis score; the tie tolerance
    # of the id rule is that gap (absolute) — with a tolerance wider than the list's own spread the rule would be one group
    gap = 0.0
    for q in range(nq):
        where = {x: i for i, x in enumerate(dec[q])}
        gap = max([gap] + [abs(got_scores[q, p] - rs2[q, where[x]]) for p, x in enumerate(out["clusters"][q]) if x in where])
    assert gap <= 5e-3, f"stage-1 scores of the GPU and the emulation differ by {gap:.2e} on shared hypotheses"
    tie = max(gap, 2e-4)
    moved = foreign = shared = 0
    sizes = []
    why = []
    for q in range(nq):
        def explain(name, q=q):        # a cluster id the emulation's final list lacks: which cut of ITS search did it fall at, and by how much?
            row = [0] + codec_ref.encode_single_newid(name, kary=V)
            w = beam_cut_explains_absence(trace, ptrace, q, R, cfg.decode_vocab_size, row, tie, final_cut=rs2[q, -1])
            why.append((q, name, w))
            return w
        m, f, sz = hypothesis_lists_match(dec[q], rs2[q], out["clusters"][q], tie, explain_foreign=explain)   # raises when two non-tied hypotheses swap
        moved, foreign, shared = moved + m, foreign + f, shared + len(set(dec[q]) & set(out["clusters"][q]))
        sizes.append(sz)
    assert shared >= 0.95 * nq * R, (shared, moved, foreign)
    assert max(max(sz) for sz in sizes) < R, "the tie rule must not degenerate into one group"
    print(f"C5 stage 1 ({nq} queries): score gap {gap:.2e} -> tie window {2 * tie:.2e}; {shared}/{nq * R} ids shared with the "
          f"emulation, {moved} moved inside a tie group, {foreign} outside the emulation's list, each explained by a tie at a cut "
          f"of the emulation's own search: {why}; tie-group sizes: {sizes}")
    # ---- stage 2 on all 64 queries: oracle rerank on the bf16-rounded rows the GPU gathered
    q_emb = state["enc_h"][:, 0].cpu()
    cand = [[m for s_ in row for m in range(int(offsets[look[s_]]), int(offsets[look[s_] + 1]))] for row in out["clusters"]]
    flat = torch.tensor(sorted({m for c in cand for m in c}), dtype=torch.long)
    rows16 = D16[flat.to(dev)].float().cpu()                   # only the candidate rows leave the GPU
    remap = {int(m): j for j, m in enumerate(flat.tolist())}
    differing = 0
    for b in range(B):
        mem_local = [remap[m] for m in cand[b]]
        ref = retrieval_ref.rerank(q_emb[b:b + 1], rows16, [mem_local], [[12] * R], [got_scores[b].astype(np.float32).tolist()],
                                   args.score_rate, R)[0]
        for a in range(len(args.score_rate)):
            ref_ids = [str(int(flat[j])) for j in ref[a][1].tolist()]
            differing += ranked_lists_match(ref_ids, ref[a][0].numpy(), out["doc_ids"][b][a], TOL)
            np.testing.assert_allclose(out["rerank_values"][b, a].cpu().numpy(), ref[a][0].numpy(), rtol=1e-4, atol=1e-4)
    assert differing <= 8, differing
    assert D16.dtype == torch.bfloat16 and D16.shape == (N, cfg.d_model)
    # ---- C5's OWN layout (BASELINE.json: "8xMI355X"; SURVEY §8e "for GDR mode, whole clusters"): the 1M-row bf16 corpus in
    # 8 cluster-aligned row shards, looped on this GPU: per shard the GDR_RERANK_POSITIONS lists of ALL 64 queries over the
    # rows [lo, hi) only -> wire form -> gdr_topk_merge_packed -> positions back to doc ids == the unsharded lists, bit for bit
    from gdr_amd.dist import shard_bounds
    A = len(args.score_rate)
    dci = retr._device_index()
    _cl, offs, cids, stride = dci.candidates(state["ids"], B, R)
    q_dev = state["enc_h"][:, 0].contiguous()
    beam32 = state["scores"].to(torch.float32).view(B, R)
    uv, ui = ops.rerank_topk(q_dev, D16, offs, cids, beam32, args.score_rate, R, max_cand=stride, cand_stride=stride)
    assert torch.equal(uv, out["rerank_values"])
    assert [[[str(x) for x in ui[b, a].tolist()] for a in range(A)] for b in range(B)] == out["doc_ids"]
    lists, rows_seen = [], 0
    for g in range(8):
        lo, hi = shard_bounds(N, 8, g, cluster_size=12)
        rows_seen += hi - lo
        v, pos = ops.rerank_topk(q_dev, D16[lo:hi], offs, cids, beam32, args.score_rate, R, max_cand=stride, cand_stride=stride,
                                 doc_range=(lo, hi), positions=True)
        lists.append(ops.topk_pack(v.view(B * A, R), pos.view(B * A, R)))
    assert rows_seen == N
    mv, mp = ops.topk_merge_packed(torch.stack(lists))
    mids = ops.rerank_positions_to_ids(mp.view(B, A * R), cids).view(B, A, R)
    assert torch.equal(mv.view(B, A, R), uv) and torch.equal(mids, ui), "8 shards merged != unsharded on the 1M bf16 corpus"
    # the exchange row of the sharded path: pack -> unpack is the identity
    w = ops.rerank_wire_pack(q_dev, beam32, offs, cids)
    q2, b2, o2, i2 = ops.rerank_wire_unpack(w, cfg.d_model, R, stride)
    live = torch.arange(stride, device=dev)[None, :] < offs[:, R:R + 1]
    assert torch.equal(q2, q_dev) and torch.equal(b2, beam32) and torch.equal(o2, offs) and torch.equal(i2[live], cids[live])


def _lightning_ckpt(t5_sd, bert_sd):
    """The layout main.py:121-126 loads: {"state_dict": {...}} with the T5 under `model.` (main_models.py:794) and the doc
    tower under `encoder.model.` (main_models.py:797, :62-78), plus entries a real checkpoint also carries."""
    sd = {"model." + k: v for k, v in t5_sd.items()}
    sd.update({"encoder.model." + k: v for k, v in bert_sd.items()})
    return {"state_dict": sd, "epoch": 3, "global_step": 1234}


def test_lightning_checkpoint_roundtrip_through_product_classes(dev, tmp_path):
    """A Lightning-style checkpoint saved with torch.save and loaded as main.py does: GDRModel (T5 part) and
    EncoderModel.from_state_dict (doc tower) give bit-identical outputs to the models built from the bare state_dicts,
    and the T5 part matches the oracle."""
    from gdr_amd.modeling import EncoderModel, GDRModel
    from oracle import beam_ref
    cfg = GDRConfig.tiny()
    sd = synth.make_state_dict(cfg, seed=77)
    bc = synth.bert_config(tiny=True)
    bsd = synth.make_bert_state_dict(bc, seed=78)
    path = str(tmp_path / "epoch=3.ckpt")
    torch.save(_lightning_ckpt(sd, bsd), path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    ids, mask = synth.make_tokens(3, L=9, vocab_hi=cfg.vocab_size, seed=2, min_len=2)
    idt, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    kw = dict(attention_mask=mt, max_length=cfg.max_output_length, num_beams=4, length_penalty=0.8, num_return_sequences=4,
              output_scores=True)
    (d1, s1), _ = GDRModel(cfg, ckpt, dev).generate(idt, **kw)
    (d0, s0), _ = GDRModel(cfg, sd, dev).generate(idt, **kw)
    assert torch.equal(d1, d0) and s1 == s0
    (rd, rs), _ = beam_ref.generate(sd, cfg, torch.from_numpy(ids), torch.from_numpy(mask), 4, restricted_head=True)
    assert np.array_equal(d1.cpu().numpy(), rd.numpy())
    np.testing.assert_allclose(np.array(s1), np.array(rs), rtol=1e-4, atol=1e-4)
    pids, pmask = synth.make_tokens(4, L=20, vocab_hi=bc["vocab_size"], seed=3, min_len=5)
    psg = {"input_ids": torch.from_numpy(pids).to(dev), "attention_mask": torch.from_numpy(pmask).to(dev)}
    e1 = EncoderModel.from_state_dict(bc, ckpt, dev)(passage=psg)
    e0 = EncoderModel.from_state_dict(bc, bsd, dev)(passage=psg)
    assert torch.equal(e1, e0)


def test_main_eval_with_infer_ckpt_equals_synthetic_weights(tmp_path):
    """`--infer_ckpt file.ckpt` (a Lightning checkpoint of the same seeded weights, t5-small sizes) gives the same res1
    TSV as the synthetic-weights run: the checkpoint path of the entry point loads what it is given."""
    from gdr_amd.main import parsers_parser
    base = [a if a != "base" else "small" for a in INFER_SH] + ["--num_return_sequences", "6", "--eval_batch_size", "3",
                                                               "--corpus_rows", "6000", "--n_queries", "5",
                                                               "--is_train_encoder", "0"]
    args = parsers_parser(base + ["--infer_ckpt", ""])
    cfg = GDRConfig.from_args(args)
    sd = synth.make_state_dict(cfg, seed=1234)
    path = str(tmp_path / "small.ckpt")
    torch.save(_lightning_ckpt(sd, {}), path)
    a, b = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    _run_main(base + ["--infer_ckpt", path, "--res1_save_path", a])
    _run_main(base + ["--infer_ckpt", "", "--res1_save_path", b])
    assert open(a).read() == open(b).read() and len(_read_tsv(a)) == 5
