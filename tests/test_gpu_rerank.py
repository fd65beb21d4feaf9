"""Stage 2 of GDR on the GPU (round 3): the device-side decode_token -> id_mapping -> candidate lists
(gdr_cluster_candidates; main_models.py:1398,1441-1443), the two-kernel in-cluster rerank in both candidate layouts, at
infer.sh's beam width (100 beams, ~1 200 candidates, k = 100; main_models.py:1574-1637, infer.sh:10-15), over a bf16 corpus
(BASELINE config C5) and over a row-sharded corpus (GDR mode) — against the oracle and against each other."""
import numpy as np
import pytest
import torch

from conftest import ranked_lists_match
from gdr_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _index(N, csz, V, holes=0):
    from gdr_amd import codec
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=csz, V=V)
    members = np.random.Generator(np.random.PCG64(5)).permutation(N).astype(np.int32)     # members are not arange
    return codec.ClusterIndex(names, offsets, members), names, depth


def _rows_for(names, picks, V, max_length, rng):
    """generate()-shaped rows [len(picks), max_length]: START, tokens of the picked name, EOS, PAD — or special cases."""
    from gdr_amd import codec
    rows = np.zeros((len(picks), max_length), np.int64)
    for r, p in enumerate(picks):
        if p == "noeos":                                      # a full-length row without EOS: decoded whole, START included
            rows[r, 1:] = [(i - 1) * V + 2 + int(rng.integers(0, V)) for i in range(1, max_length)]
        elif p == "unknown":                                  # a well-formed id that names no cluster
            toks = codec.encode_single_newid("-".join(["0"] * 7), kary=V)
            rows[r, 1:1 + len(toks)] = toks
        elif p == "empty":                                    # START, EOS: the empty string
            rows[r, 1] = 1
        else:
            toks = codec.encode_single_newid(names[p], kary=V)
            rows[r, 1:1 + len(toks)] = toks
    return rows


def test_cluster_candidates_device_equals_host_lookup(dev):
    """gdr_cluster_candidates against codec.decode_token + ClusterIndex.candidates (the reference's string path) on rows
    that name clusters, repeat one, name none, carry no EOS, or are empty; plus a cluster whose name is the no-EOS garbage
    string itself (what the C3 test does to give random-weight decodes real candidates)."""
    from gdr_amd import codec, ops
    V, ml, B, R = 30, 10, 9, 6
    index, names, depth = _index(7003, 12, V)
    rng = np.random.Generator(np.random.PCG64(1))
    picks = [int(rng.integers(0, len(names))) for _ in range(B * R)]
    picks[3] = picks[2]                                       # the same cluster twice in one query
    picks[7], picks[13], picks[20] = "unknown", "noeos", "empty"
    picks[R * 4:R * 5] = ["unknown"] * R                      # a query without any candidate
    rows = _rows_for(names, picks, V, ml, rng)
    # rename one cluster to the string a no-EOS row decodes to, and one to the empty string
    garbage = codec.decode_token(rows[13:14], kary=V, output_vocab_size=V)[0]
    renamed = list(names)
    renamed[5], renamed[6] = garbage, ""
    index = codec.ClusterIndex(renamed, index.offsets, index.members)
    dec = codec.dec_2d(codec.decode_token(rows, kary=V, output_vocab_size=V), R)
    offs_h, ids_h, max_h = index.candidates(dec)
    dci = ops.DeviceClusterIndex(index, dev, V)
    cl, offs, ids, stride = dci.candidates(torch.from_numpy(rows).to(dev), B, R)
    assert stride == R * 12 and max_h <= stride
    offs, ids, cl = offs.cpu().numpy(), ids.cpu().numpy(), cl.cpu().numpy()
    offs_h, ids_h = offs_h.numpy(), ids_h.numpy()
    for b in range(B):
        base = offs_h[b * R]
        assert np.array_equal(offs[b], offs_h[b * R:(b + 1) * R + 1] - base), b
        n = offs[b, R]
        assert np.array_equal(ids[b, :n], ids_h[base:base + n]), b
    want_cl = [index.lookup.get(s, -1) for row in dec for s in row]
    assert cl.tolist() == want_cl and cl[13] == 5 and cl[20] == 6 and cl[7] == -1 and (cl[R * 4:R * 5] == -1).all()


@pytest.mark.parametrize("B,R,csz,k", [(64, 10, 12, 10), (2, 100, 12, 100), (1, 100, 12, 100), (3, 100, 30, 100)])
def test_rerank_at_c3_and_infer_sh_widths_vs_oracle(dev, B, R, csz, k):
    """The rerank at C3's shape (64 x 10 beams x 12-doc clusters) and at infer.sh's (100 beams: ~1 200 candidates, k = 100,
    the 2048-key sort; 30-doc clusters: 3 000 candidates, 4096 keys) from device-built candidate blocks, against the oracle
    for every alpha; and the block layout against the one-CSR layout of the reference's concatenation (same bits)."""
    from gdr_amd import codec, ops
    from oracle import retrieval_ref
    V, ml, N, d = 30, 10, 60000, 768
    index, names, depth = _index(N, csz, V)
    rng = np.random.Generator(np.random.PCG64(B * 1000 + R))
    picks = rng.choice(len(names), size=(B, R), replace=True).reshape(-1).tolist()
    picks[1] = "unknown"
    rows = _rows_for(names, picks, V, ml, rng)
    D = synth.make_corpus(N, d, seed=3)
    Q, _ = synth.make_queries(D, B, seed=4)
    Q *= 0.15                                                   # away from tanh saturation
    beam = np.sort(rng.standard_normal((B, R)).astype(np.float32) * 2 - 8, axis=1)[:, ::-1].copy()
    alphas = [0, 0.5, 1, 1.5, 2, 2.5, 3]
    dci = ops.DeviceClusterIndex(index, dev, V)
    Qd, Dd, bd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev), torch.from_numpy(beam).to(dev)
    cl, offs, ids, stride = dci.candidates(torch.from_numpy(rows).to(dev), B, R)
    v, i = ops.rerank_topk(Qd, Dd, offs, ids, bd, alphas, k, max_cand=stride, cand_stride=stride)
    # the reference's layout: one CSR
    dec = codec.dec_2d(codec.decode_token(rows, kary=V, output_vocab_size=V), R)
    offs_h, ids_h, max_h = index.candidates(dec)
    v2, i2 = ops.rerank_topk(Qd, Dd, offs_h.to(dev), ids_h.to(dev), bd, alphas, k, max_cand=max_h)
    assert torch.equal(v, v2) and torch.equal(i, i2)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    sizes = np.diff(index.offsets)
    for b in range(B):
        mem = [m for s in dec[b] for m in index[s]]
        num = [len(index[s]) for s in dec[b]]
        ref = retrieval_ref.rerank(torch.from_numpy(Q[b:b + 1]), torch.from_numpy(D), [mem], [num], beam[b:b + 1].tolist(),
                                   alphas, min(k, len(mem)))[0]
        for a in range(len(alphas)):
            rv, ri = ref[a]
            kk = rv.numel()
            np.testing.assert_allclose(v[b, a, :kk], rv.numpy(), rtol=TOL, atol=TOL)
            ranked_lists_match(ri.tolist(), rv.numpy(), i[b, a, :kk].tolist(), TOL)
            assert (i[b, a, kk:] == -1).all()
    assert int(sizes.max()) == csz


def test_rerank_bf16_corpus_vs_oracle_on_rounded_rows(dev):
    """gdr_rerank_topk_bf16 (config C5): the corpus stays bf16 (rows gathered as bf16, widened exactly), the dot is the
    fp32 chain against the fp32 query — so it must equal the oracle applied to the bf16-ROUNDED corpus at fp32 tolerance,
    ids exact outside tolerance-tie groups, and differ from the unrounded-corpus result only at bf16 tolerance."""
    from gdr_amd import codec, ops
    from oracle import retrieval_ref
    V, ml, N, d, B, R, k = 30, 10, 50000, 768, 16, 30, 30
    index, names, depth = _index(N, 12, V)
    rng = np.random.Generator(np.random.PCG64(9))
    rows = _rows_for(names, rng.choice(len(names), size=B * R).tolist(), V, ml, rng)
    D = synth.make_corpus(N, d, seed=13)
    Q, _ = synth.make_queries(D, B, seed=14)
    Q *= 0.15
    beam = np.sort(rng.standard_normal((B, R)).astype(np.float32) - 6, axis=1)[:, ::-1].copy()
    alphas = [0, 1, 3]
    Dd = torch.from_numpy(D).to(dev)
    D16 = ops.to_bf16(Dd)
    assert D16.dtype == torch.bfloat16
    dci = ops.DeviceClusterIndex(index, dev, V)
    cl, offs, ids, stride = dci.candidates(torch.from_numpy(rows).to(dev), B, R)
    Qd, bd = torch.from_numpy(Q).to(dev), torch.from_numpy(beam).to(dev)
    before = torch.cuda.memory_allocated(dev)
    v, i = ops.rerank_topk(Qd, D16, offs, ids, bd, alphas, k, max_cand=stride, cand_stride=stride)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) - before < (8 << 20), "the bf16 corpus must not be up-cast"
    v32, i32 = ops.rerank_topk(Qd, Dd, offs, ids, bd, alphas, k, max_cand=stride, cand_stride=stride)
    np.testing.assert_allclose(v.cpu().numpy(), v32.cpu().numpy(), rtol=2e-2, atol=2e-2)
    Dr = D16.float().cpu()                                       # the rounded corpus, as fp32
    dec = codec.dec_2d(codec.decode_token(rows, kary=V, output_vocab_size=V), R)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    differing = 0
    for b in range(B):
        mem = [m for s in dec[b] for m in index[s]]
        num = [len(index[s]) for s in dec[b]]
        ref = retrieval_ref.rerank(torch.from_numpy(Q[b:b + 1]), Dr, [mem], [num], beam[b:b + 1].tolist(), alphas, k)[0]
        for a in range(len(alphas)):
            rv, ri = ref[a]
            np.testing.assert_allclose(v[b, a], rv.numpy(), rtol=TOL, atol=TOL)
            differing += ranked_lists_match(ri.tolist(), rv.numpy(), i[b, a].tolist(), TOL)   # asserts outside tie groups
    assert differing <= 4, differing                             # permutations inside fp32 tolerance ties are rare


@pytest.mark.parametrize("bf16", [False, True])
def test_rerank_8_shards_merged_is_bit_identical_to_unsharded(dev, bf16):
    """GDR mode on a row-sharded corpus, on one GPU: 8 cluster-aligned shards, each reranks the candidates whose doc ids
    fall in its [lo, hi) and emits per-(query, alpha) {score, position} lists; packed (gdr_topk_pack), merged over B*A rows
    (gdr_topk_merge_packed) and mapped back to doc ids they must be BIT-IDENTICAL to the unsharded gdr_rerank_topk — a
    candidate's score does not depend on the shard.  Includes a query whose candidates all live in one shard, one with
    fewer than k candidates, and members that are not sorted by id (positions, not ids, break ties)."""
    from gdr_amd import codec, ops
    from gdr_amd.dist import shard_bounds
    V, ml, N, d, B, R, k, G, csz = 30, 10, 40003, 768, 12, 10, 25, 8, 12
    names, depth, offsets, _m = synth.make_cluster_ids(N, cluster_size=csz, V=V)
    members = np.arange(N, dtype=np.int32)
    rng = np.random.Generator(np.random.PCG64(21))
    for c in range(len(names)):                                  # shuffle inside clusters: a cluster still owns a contiguous id range
        lo, hi = offsets[c], offsets[c + 1]
        members[lo:hi] = rng.permutation(members[lo:hi])
    index = codec.ClusterIndex(names, offsets, members)
    picks = rng.choice(len(names), size=(B, R)).tolist()
    picks[0][:3] = [8, 9, 1666]                                  # clusters made of exact duplicates (below), two shards
    picks[1] = list(range(20, 20 + R))                           # all candidates of query 1 live in shard 0
    picks[2] = [7] + ["unknown"] * (R - 1)                       # 12 candidates < k
    rows = _rows_for(names, [p for row in picks for p in row], V, ml, rng)
    D = synth.make_corpus(N, d, seed=31)
    D[100:120] = D[100]                                          # exact duplicates: exactly tied scores inside one query,
    D[19992:20004] = D[100]                                      # ... and across shards (alpha = 0: every copy ties)
    Q, _ = synth.make_queries(D, B, seed=32)
    Q *= 0.15
    beam = np.sort(rng.standard_normal((B, R)).astype(np.float32) - 6, axis=1)[:, ::-1].copy()
    alphas = [0, 0.5, 2]
    A = len(alphas)
    Dd = torch.from_numpy(D).to(dev)
    if bf16:
        Dd = ops.to_bf16(Dd)
    Qd, bd = torch.from_numpy(Q).to(dev), torch.from_numpy(beam).to(dev)
    dci = ops.DeviceClusterIndex(index, dev, V)
    cl, offs, ids, stride = dci.candidates(torch.from_numpy(rows).to(dev), B, R)
    v0, i0 = ops.rerank_topk(Qd, Dd, offs, ids, bd, alphas, k, max_cand=stride, cand_stride=stride)
    packs, seen = [], 0
    for g in range(G):
        lo, hi = shard_bounds(N, G, g, cluster_size=csz)
        vg, pg = ops.rerank_topk(Qd, Dd[lo:hi], offs, ids, bd, alphas, k, max_cand=stride, cand_stride=stride,
                                 doc_range=(lo, hi), positions=True)
        seen += int((pg >= 0).sum())
        packs.append(ops.topk_pack(vg.view(B * A, k), pg.view(B * A, k)))
    mv, mp = ops.topk_merge_packed(torch.stack(packs))
    mp = mp.view(B, A * k).long()
    mi = torch.where(mp >= 0, ids.long().gather(1, mp.clamp(min=0)), mp).view(B, A, k).to(torch.int32)
    assert torch.equal(mv.view(B, A, k), v0) and torch.equal(mi, i0)
    assert int((i0[2] >= 0).sum()) == A * 12 and seen >= int((i0 >= 0).sum())
    sh1 = [g for g in range(G) if shard_bounds(N, G, g, csz)[0] <= int(i0[1, 0, 0]) < shard_bounds(N, G, g, csz)[1]]
    assert sh1 == [0]


def test_l2_normalize_vs_torch(dev):
    """gdr_l2_normalize = torch.nn.functional.normalize(dim=-1) (DensePooler, dense.py:24-25), incl. an all-zero row."""
    from gdr_amd import ops
    x = torch.randn(37, 768, generator=torch.Generator().manual_seed(3))
    x[5] = 0
    got = ops.l2_normalize(x.to(dev)).cpu()
    torch.testing.assert_close(got, torch.nn.functional.normalize(x, dim=-1), rtol=1e-6, atol=1e-7)
    assert torch.equal(got[5], torch.zeros(768))


@pytest.mark.parametrize("d", [768, 64, 1024, 2052])
def test_t5_layer_norm_divides_like_the_reference_bit_for_bit(dev, d):
    """T5LayerNorm is `weight * (x / sqrt(mean(x^2) + eps))` (modeling_t5.py:164-171): a true division per element.  The kernels
    divide a row by its one denominator in three instructions per element (correctly rounded reciprocal, q = x * r, exact
    residual, one correction — common.h RowDivisor) and claim the IEEE quotient's bits.  Checked where the expected bits do not
    depend on a summation order: rows of integers (every x^2 and every partial sum exact in fp32), scaled per row by a power of
    two, so that sum / d, + eps, sqrt and the division are each ONE correctly rounded fp32 operation — exactly what numpy's
    float32 arithmetic computes.  200 000 rows = 200 000 different denominators; d = 2052 takes the kernel's second form
    (rows wider than its registers)."""
    from gdr_amd import ops
    rng = np.random.default_rng(17 + d)
    rows = 200000 if d <= 1024 else 20000
    hi = int(np.sqrt((1 << 24) / d))                                   # d * hi^2 < 2^24: the sum of squares stays exact
    xi = rng.integers(-hi, hi + 1, size=(rows, d)).astype(np.float32)
    scale = np.exp2(rng.integers(-6, 7, size=(rows, 1))).astype(np.float32)
    x = xi * scale                                                      # exact (power of two)
    w = rng.standard_normal(d).astype(np.float32)
    eps = np.float32(1e-6)
    ss = (xi.astype(np.int64) ** 2).sum(1, keepdims=True).astype(np.float32) * (scale * scale)   # exact, any order
    denom = np.sqrt(ss / np.float32(d) + eps, dtype=np.float32)
    want = w[None, :] * (x / denom)
    got = ops.t5_layer_norm(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), float(eps)).cpu().numpy()
    assert got.dtype == np.float32 and want.dtype == np.float32
    bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(1))
    assert bad.size == 0, f"{bad.size} of {rows} rows differ from the IEEE expression, first: row {bad[0]}"


def test_one_oversized_cluster_does_not_break_the_step(dev):
    """The device candidate blocks are `num_beams x largest cluster of the corpus` wide.  One outlier cluster (here 1 000 docs
    at 10 beams: 10 000 > the rerank's 8 192-candidate cap) must not make every step fail when the clusters actually decoded
    are small (the host CSR path sized its bound from the data): ops.block_max_cand then reads the real per-query counts — and
    a query that REALLY decodes more than 8 192 candidates is refused with a clear error, as before."""
    from gdr_amd import _ffi, codec, ops
    from oracle import retrieval_ref
    V, ml, d, R, B = 30, 10, 64, 10, 4
    sizes = np.full(300, 12, np.int64)
    sizes[7] = 1000                                            # the outlier
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    N = int(offsets[-1])
    names = ["-".join(str(x) for x in synth.cluster_digits(c, 2, V)) for c in range(300)]
    index = codec.ClusterIndex(names, offsets, np.arange(N, dtype=np.int32))
    rng = np.random.Generator(np.random.PCG64(3))
    picks = [int(x) for x in rng.choice([c for c in range(300) if c != 7], size=B * R)]
    rows = _rows_for(names, picks, V, ml, rng)
    D = synth.make_corpus(N, d, seed=3)
    Q, _ = synth.make_queries(D, B, seed=4)
    Q *= 0.3
    beam = np.sort(rng.standard_normal((B, R)).astype(np.float32), axis=1)[:, ::-1].copy()
    dci = ops.DeviceClusterIndex(index, dev, V)
    Qd, Dd, bd = torch.from_numpy(Q).to(dev), torch.from_numpy(D).to(dev), torch.from_numpy(beam).to(dev)
    cl, offs, ids, stride = dci.candidates(torch.from_numpy(rows).to(dev), B, R)
    assert stride == R * 1000 > ops.RERANK_MAX_CAND
    mc = ops.block_max_cand(offs, R, stride)
    assert mc == R * 12
    v, i = ops.rerank_topk(Qd, Dd, offs, ids, bd, [0, 1.5], R, max_cand=mc, cand_stride=stride)
    dec = codec.dec_2d(codec.decode_token(rows, kary=V, output_vocab_size=V), R)
    mem = [[m for s_ in row for m in index[s_]] for row in dec]
    num = [[len(index[s_]) for s_ in row] for row in dec]
    ref = retrieval_ref.rerank(torch.from_numpy(Q), torch.from_numpy(D), mem, num, beam.tolist(), [0, 1.5], R)
    for b in range(B):
        for a in range(2):
            np.testing.assert_allclose(v[b, a].cpu().numpy(), ref[b][a][0].numpy(), rtol=TOL, atol=TOL)
            ranked_lists_match(ref[b][a][1].tolist(), ref[b][a][0].numpy(), i[b, a].cpu().tolist(), TOL)
    # a query that really decodes the outlier ten times: 10 000 candidates
    rows2 = _rows_for(names, [7] * (B * R), V, ml, rng)
    _c, offs2, _i, _s = dci.candidates(torch.from_numpy(rows2).to(dev), B, R)
    with pytest.raises(_ffi.GdrError, match="at most 8192"):
        ops.block_max_cand(offs2, R, stride)
