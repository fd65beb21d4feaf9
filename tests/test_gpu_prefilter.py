"""gdr_sim_topk_prefilter: the fp32 top-k through a bf16 pre-filter (include/gdr_hip.h; call site dense.py:53-54 + topk as at
main_models.py:1625).  The contract is exactness: the result is the top-k of the fp32 scores for EVERY input — the bf16 pass only
decides which docs get an fp32 score — so the tests compare with the all-fp32 path (gdr_sim_topk) at a tolerance that only covers
the two fp32 summation orders (1e-6), on benign and on adversarial corpora, and with the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import order_insensitive_topk_match
from gdr_amd import synth

pytestmark = pytest.mark.gpu
TOL32 = 2e-6          # two fp32 summation orders of the same 768 products


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.set_grad_enabled(False)
    return torch.device("cuda:0")


def _both(Q, D, k, dev, **kw):
    from gdr_amd import ops
    Qd, Dd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev)
    P = ops.PrefilteredCorpus(Dd)
    fv, fi, fs = ops.sim_topk(Qd, Dd, k, return_status=True, **kw)
    pv, pi, ps = ops.sim_topk(Qd, P, k, return_status=True, **kw)
    return (fv.cpu().numpy(), fi.cpu().numpy().astype(np.int64), fs.cpu().numpy()), (pv.cpu().numpy(), pi.cpu().numpy().astype(np.int64), ps.cpu().numpy()), P


def test_prefilter_c2_shape_all_512_rows_equal_the_fp32_path_and_the_oracle(dev):
    """BASELINE config C2's similarity (512 queries x 320 000 x 768, top-100): every row against the all-fp32 path under the top-k
    rule at 2e-6, and eight rows against the CPU oracle at the usual 1e-4."""
    from oracle import retrieval_ref
    N, B, k = 320000, 512, 100
    D = synth.make_corpus(N, 768)
    Q, _ = synth.make_queries(D, B)
    (fv, fi, fs), (pv, pi, ps), P = _both(Q, D, k, dev)
    assert fs.sum() == 0 and ps.sum() == 0
    assert abs(P.dnorm_max - float(np.linalg.norm(D, axis=1).max())) < 1e-4
    permuted = order_insensitive_topk_match(fv, fi, pv, pi, TOL32)
    assert (fi == pi).all(axis=1).mean() > 0.97, "ids differ from the fp32 path in more than 3 % of the rows"   # fp32-noise ties only
    assert permuted <= 40
    rows = np.linspace(0, B - 1, 8).astype(np.int64)
    rv, ri = retrieval_ref.sim_topk(torch.from_numpy(Q[rows]), torch.from_numpy(D), k)
    order_insensitive_topk_match(rv.numpy(), ri.numpy(), pv[rows], pi[rows], 1e-4)


def test_prefilter_is_exact_where_bf16_cannot_tell_docs_apart(dev):
    """Adversarial for the pre-filter: 400 docs that are ONE vector plus perturbations of 1e-5 — identical bf16 images, different fp32
    scores — in a corpus of ordinary docs, queried along that vector with k = 100: the bf16 pass sees 400 equal scores, the answer is
    the 100 best by fp32.  Plus 150 exact duplicates of another vector (ties at every rank: lower id first) and k cutting through them."""
    rng = np.random.default_rng(5)
    N, d, k = 50000, 768, 100
    D = synth.make_corpus(N, d, seed=3)
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    where = rng.choice(N, 400, replace=False)
    D[where] = base[None, :] + 1e-5 * rng.standard_normal((400, d)).astype(np.float32)
    dup = rng.standard_normal(d).astype(np.float32)
    dup /= np.linalg.norm(dup)
    where2 = np.setdiff1d(rng.choice(N, 170, replace=False), where)[:150]
    D[where2] = dup[None, :]
    Q = np.stack([base, dup] + [base * 0.7 + dup * 0.7] * 2 + [rng.standard_normal(d).astype(np.float32) for _ in range(60)]).astype(np.float32)
    (fv, fi, fs), (pv, pi, ps), _ = _both(Q, D, k, dev)
    assert fs.sum() == 0 and ps.sum() == 0
    order_insensitive_topk_match(fv, fi, pv, pi, TOL32)
    assert set(pi[0].tolist()) <= set(where.tolist()), "query 0's top-100 are perturbed copies of its own vector"
    np.testing.assert_array_equal(pi[1], np.sort(where2)[:k])          # exact ties: lower doc id first, as torch.topk / the fp32 path
    np.testing.assert_array_equal(fi[1], pi[1])
    # against float64 ground truth for query 0: every doc that beats the 100th score by more than fp32 noise is returned, nothing that
    # loses by more than that is (the 400 scores are spread over ~6e-5 around 1.0: ~1.5e-7 = one fp32 ulp apart on average)
    s64 = D.astype(np.float64) @ Q[0].astype(np.float64)
    cut = np.sort(s64)[-k]
    must = set(np.nonzero(s64 > cut + 2e-6)[0].tolist())
    may = set(np.nonzero(s64 >= cut - 2e-6)[0].tolist())
    for name, got in (("fp32 path", fi[0]), ("pre-filter", pi[0])):
        got = set(got.tolist())
        assert must <= got <= may, (name, len(must - got), len(got - may))
    assert len(must) > 60 and len(may) < 140


def test_prefilter_band_overflow_falls_back_to_the_fp32_path(dev):
    """More docs inside the 2-eps band than the rescoring list holds (3 000 exact duplicates, cap 1 024): status flags the query and
    ops.sim_topk recomputes it on the fp32 path — the answer stays exact."""
    from gdr_amd import ops
    rng = np.random.default_rng(7)
    N, d, k = 40000, 768, 50
    D = synth.make_corpus(N, d, seed=4)
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    where = np.sort(rng.choice(N, 3000, replace=False))
    D[where] = v[None, :]
    Q = np.stack([v] + [rng.standard_normal(d).astype(np.float32) for _ in range(63)]).astype(np.float32)
    Qd, Dd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev)
    P = ops.PrefilteredCorpus(Dd)
    _, _, st = ops.sim_topk(Qd, P, k, return_status=True, exact_on_overflow=False)
    assert st.cpu().numpy()[0] == 1 and st.cpu().numpy()[1:].sum() == 0
    pv, pi = ops.sim_topk(Qd, P, k)                                      # exact_on_overflow=True: the flagged query is redone
    np.testing.assert_array_equal(pi[0].cpu().numpy(), where[:k])
    fv, fi = ops.sim_topk(Qd, Dd, k)
    order_insensitive_topk_match(fv.cpu().numpy(), fi.cpu().numpy().astype(np.int64), pv.cpu().numpy(), pi.cpu().numpy().astype(np.int64), TOL32)


@pytest.mark.parametrize("N,B,k,scale", [(20001, 64, 10, 1.0), (9000, 40, 1, 37.0), (130000, 100, 300, 0.01), (16500, 33, 1000, 5.0)])
def test_prefilter_shapes_and_scales(dev, N, B, k, scale):
    """Ragged corpus sizes (incl. the small-corpus plan without a filter pass), k = 1 .. 1000, un-normalised embeddings (the band
    scales with ||q|| * max ||d||), negative scores."""
    rng = np.random.default_rng(N)
    D = (synth.make_corpus(N, 768, seed=N % 97) * scale * (0.5 + rng.random((N, 1)))).astype(np.float32)
    Q = (rng.standard_normal((B, 768)) * scale * 0.3).astype(np.float32)
    Q[0] = -np.abs(Q[0])                                                  # a query whose best scores are negative for many docs
    (fv, fi, fs), (pv, pi, ps), _ = _both(Q, D, k, dev)
    assert fs.sum() == 0 and ps.sum() == 0
    tol = TOL32 * max(1.0, scale * scale)
    order_insensitive_topk_match(fv, fi, pv, pi, tol)


@pytest.mark.parametrize("B", [1, 7, 32])
def test_latency_mode_bf16_stream_and_prefilter(dev, B):
    """B <= 32 (latency mode): the corpus-wide pass of the pre-filter is sim_stream_bf16_kernel — the HBM stream over the bf16 image,
    half the bytes of the fp32 stream.  (a) the bf16-corpus entry point (gdr_sim_topk_bf16) on it equals the oracle on the bf16-rounded
    operands; (b) the pre-filtered fp32 result equals the all-fp32 path under the top-k rule; ragged N (not a multiple of 128)."""
    from gdr_amd import ops
    from oracle import retrieval_ref
    N, k = 100037, 100
    D = synth.make_corpus(N, 768, seed=11)
    Q, _ = synth.make_queries(D, B, seed=12)
    Qd, Dd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev)
    D16 = ops.to_bf16(Dd)
    bv, bi = ops.sim_topk(Qd, D16, k)
    rv, ri = retrieval_ref.sim_topk(torch.from_numpy(Q).bfloat16().float(), torch.from_numpy(D).bfloat16().float(), k)
    order_insensitive_topk_match(rv.numpy(), ri.numpy(), bv.cpu().numpy(), bi.cpu().numpy().astype(np.int64), 1e-5)
    fv, fi = ops.sim_topk(Qd, Dd, k)
    pv, pi = ops.sim_topk(Qd, ops.PrefilteredCorpus(Dd), k)
    order_insensitive_topk_match(fv.cpu().numpy(), fi.cpu().numpy().astype(np.int64), pv.cpu().numpy(), pi.cpu().numpy().astype(np.int64), TOL32)
    ov, oi = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), k)
    order_insensitive_topk_match(ov.numpy(), oi.numpy(), pv.cpu().numpy(), pi.cpu().numpy().astype(np.int64), 1e-4)


@pytest.mark.parametrize("B", [4, 40])
def test_prefilter_band_holds_under_coherent_bf16_rounding(dev, B):
    """r06 (advisor finding on r05): bf16 has an 8-bit significand, so RNE's unit roundoff is 2^-8 and the per-doc bound is
    ||q||·||d||·(2^-7 + 2^-16 + ...) — r05 coded half of that.  An input whose coordinates all round the SAME way reaches 2/3 of the
    true bound: q and doc A hold 1 + 2^-8 - 2^-14 on 384 coordinates (bf16 rounds them DOWN to 1), q and doc B hold 1 + 2^-8 + 2^-14
    on the next 383 (rounded UP to 1 + 2^-7).  fp32: A = 386.96 > B = 386.04; bf16 operands: A = 384.0 < B = 389.0.  With the r05
    band (2 eps = 4.5) A — the true top-1 — fell outside [389.0 - 4.5, ...] and was dropped with status 0; the band must keep it."""
    from gdr_amd import ops
    d, N = 768, 20000
    lo, hi = np.float32(1 + 2.0 ** -8 - 2.0 ** -14), np.float32(1 + 2.0 ** -8 + 2.0 ** -14)
    # background docs of the SAME norm as A and B (||A|| = 19.7): eps_q scales with max ||d||, so tiny background docs would all fall
    # inside the band of the sample threshold and overflow the list (status 1 -> the fp32 fallback, which is not what is tested here)
    D = (synth.make_corpus(N, d, seed=21) * 19.0).astype(np.float32)
    ia, ib = 777, 12345
    D[ia] = 0
    D[ia, :384] = lo
    D[ib] = 0
    D[ib, 384:767] = hi
    q = np.zeros(d, np.float32)
    q[:384], q[384:767] = lo, hi
    Q = np.repeat(q[None], B, 0)
    Q[1::2] *= np.float32(0.5)                                            # the band scales with ||q||
    Qd, Dd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev)
    s16 = (torch.from_numpy(D[[ia, ib]]).bfloat16().float() @ torch.from_numpy(q).bfloat16().float()).numpy()
    s32 = D[[ia, ib]].astype(np.float64) @ q.astype(np.float64)
    assert s32[0] > s32[1] + 0.9 and s16[1] > s16[0] + 4.9, "the construction: fp32 ranks A first, bf16 operands rank B first by 5.0"
    for k in (1, 2, 10):
        fv, fi, fs = ops.sim_topk(Qd, Dd, k, return_status=True)
        pv, pi, ps = ops.sim_topk(Qd, ops.PrefilteredCorpus(Dd), k, return_status=True, exact_on_overflow=False)
        assert int(fs.sum()) == 0 and int(ps.sum()) == 0
        assert (fi[:, 0].cpu().numpy() == ia).all() and (pi[:, 0].cpu().numpy() == ia).all(), "the true top-1 (doc A) was filtered out"
        if k > 1:
            assert (pi[:, 1].cpu().numpy() == ib).all()
        order_insensitive_topk_match(fv.cpu().numpy(), fi.cpu().numpy().astype(np.int64), pv.cpu().numpy(),
                                     pi.cpu().numpy().astype(np.int64), 1e-4)
