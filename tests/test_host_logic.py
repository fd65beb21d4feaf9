"""CPU-only checks of the boundary and host logic: the C-ABI library loads and exports every symbol that
include/gdr_hip.h declares; host routines inside it (no GPU work) match the reference's golden vectors."""
import os
import sys
import re

import numpy as np
import pytest

from conftest import REPO, golden


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "gdr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gdr_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_bound_and_exported():
    from gdr_amd import _ffi
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_ffi.SIGNATURES) == declared, (sorted(set(declared) ^ set(_ffi.SIGNATURES)))
    l = _ffi.lib()                      # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert hasattr(l, name)
    assert l.gdr_abi_version() == _ffi.ABI_VERSION == 8


def test_missing_library_fails_loudly(monkeypatch):
    from gdr_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libgdr_hip.so")
    with pytest.raises(_ffi.GdrError):
        _ffi.lib()


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "gdr_amd")
    for root, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src or f == "synth.py", f


def test_relative_bucket_table_bit_exact_vs_reference():
    from gdr_amd import ops
    g = golden("g2_buckets")
    bi = ops.relative_bucket_table(True, 32, 128, 128, 128).numpy()
    uni = ops.relative_bucket_table(False, 32, 128, 128, 128).numpy()
    assert np.array_equal(bi, g["bidirectional"].astype(np.int32))
    assert np.array_equal(uni, g["unidirectional"].astype(np.int32))


def test_argument_errors_are_reported_without_gpu():
    """Shape validation happens before any launch, so it is testable on CPU."""
    from gdr_amd import _ffi
    l = _ffi.lib()
    rc = l.gdr_sim_topk(None, 4, None, 100, 768, 10, 0, None, None, None, 0, None, 0, None)
    assert rc == _ffi.GDR_EINVAL and b"null" in l.gdr_last_error()
    rc = l.gdr_linear_f32(None, 8, None, 8, None, 8, 4, 4, 8, 0, None, None, 0, None)
    assert rc == _ffi.GDR_EINVAL
    assert 0 < l.gdr_sim_topk_workspace_bytes(512, 320000, 768, 100, 0) < l.gdr_sim_topk_workspace_bytes(512, 320000, 768, 100, 1)


def test_product_codec_known_answers():
    import types
    from gdr_amd import codec
    g = golden("g6_codec")
    args = types.SimpleNamespace(kary=30, position=1, output_vocab_size=30)
    off = 0
    for s, n in zip(g["strs"], g["enc_len"]):
        assert codec.encode_single_newid(args, str(s)) == g["enc_flat"][off:off + n].tolist()
        assert codec.encode_single_newid(str(s), kary=30) == g["enc_flat"][off:off + n].tolist()
        off += n
    assert codec.decode_token(args, g["seqs"]) == [str(x) for x in g["dec"]]
    assert codec.encode_single_newid(types.SimpleNamespace(kary=0, position=1), "40917") == g["enc10"].tolist()
    d2 = codec.dec_2d(list(range(10)), 4)
    assert [x for r in d2 for x in r] == g["dec2d_flat"].tolist() and [len(r) for r in d2] == g["dec2d_len"].tolist()
    # round trip on every 3-digit base-30 id boundary
    for s in ["0-0-0", "29-29-29", "1-2-3-4-5-6-7-8-9"]:
        toks = np.array([[0] + codec.encode_single_newid(s, kary=30)])
        assert codec.decode_token(toks, kary=30, output_vocab_size=30) == [s]


def test_product_metrics_known_answers(tmp_path):
    import types
    from gdr_amd import codec
    g = golden("g7_metrics")
    rows = [tuple(str(x) for x in r) for r in g["rows"]]
    p = tmp_path / "res1.tsv"
    codec.write_res1(str(p), rows)
    for k, v in zip(g["recall_k"], g["recall_v"]):
        args = types.SimpleNamespace(res1_save_path=str(p), trivia=0, recall_num=[int(k)])
        assert codec.recall(args, verbose=False) == float(v)
    assert codec.MRR100(types.SimpleNamespace(res1_save_path=str(p)), verbose=False) == pytest.approx(float(g["mrr100"]), abs=1e-15)


@pytest.mark.parametrize("case", ["two_stage", "cluster_only", "single_gt"])
def test_epoch_metrics_match_reference_validation_epoch_end(case):
    """cal_recall / cal_accuracy / cal_MRR / cal_MAP and the per-alpha grouping of validation_epoch_end
    (main_models.py:1643-1908): every value the reference passed to self.log on these step outputs, by name."""
    import json
    import types
    from gdr_amd import codec
    g = golden("g12_epoch_metrics")
    outputs = json.loads(str(g[case + "_outputs"]))
    args = types.SimpleNamespace(**json.loads(str(g[case + "_args"])))
    logged = codec.validation_epoch_end(outputs, args)
    keys = [str(k) for k in g[case + "_keys"]]
    assert sorted(logged) == keys
    np.testing.assert_allclose([logged[k] for k in keys], g[case + "_vals"], rtol=0, atol=1e-12)
    assert any(0 < v < 1 for v in logged.values())                      # the fixture is not degenerate


def test_cal_metrics_small_known_answers():
    from gdr_amd import codec
    q_pred = {"a": ["1", "2", "3", "4"], "b": ["9", "8", "7", "6"]}
    q_gt = {"a": ["3", "1"], "b": ["5"]}
    assert codec.cal_recall(q_pred, q_gt, 2) == (0.25, 1 / 3)           # macro (1/2 + 0)/2, micro 1/3
    assert codec.cal_accuracy(q_pred, q_gt, 1) == 0.5
    assert codec.cal_MRR(q_pred, q_gt, 4) == 0.5
    assert codec.cal_MAP(q_pred, q_gt, 4) == pytest.approx(((1 / 1 + 2 / 3) / 4) / 2)


def test_cluster_index_csr_matches_reference_lookup():
    from gdr_amd import codec
    g = golden("g4_rerank")
    names = [str(x) for x in g["names"]]
    index = codec.ClusterIndex(names, g["offsets"], g["members"])
    B, R = g["chosen"].shape
    dec = codec.dec_2d(codec.decode_token(g["dec_ids"], output_vocab_size=6, kary=6), R)
    assert [",".join(d) for d in dec] == [str(x) for x in g["cluster_strs"]]
    offs, ids, max_cand = index.candidates(dec)
    assert offs.shape[0] == B * R + 1 and int(offs[-1]) == ids.shape[0]
    for b in range(B):
        for j in range(R):
            c = int(g["chosen"][b, j])
            seg = ids[int(offs[b * R + j]):int(offs[b * R + j + 1])].tolist()
            assert seg == g["members"][g["offsets"][c]:g["offsets"][c + 1]].tolist()
    assert index["no-such-cluster"] == [] and max_cand == max(int(offs[(b + 1) * R] - offs[b * R]) for b in range(B))
    # dict round trip (the reference pickles a dict of lists, main_models.py:874-889)
    idx2 = codec.ClusterIndex.from_id_mapping({n: index[n] for n in names})
    assert np.array_equal(idx2.offsets, index.offsets) and np.array_equal(idx2.members, index.members)


def test_cli_namespace_matches_reference_parser():
    """Same flag names, types, defaults and post-processing as GDR_model/main.py:260-448 (golden = the reference's
    own parser run in the build container); --trivia is additionally tolerated because infer.sh passes it."""
    import json
    from gdr_amd.main import parsers_parser
    g = golden("g9_cli")
    for key, argv in (("default", []), ("infer_sh", str(g["infer_argv"]).split())):
        ref = json.loads(str(g[key]))
        mine = vars(parsers_parser(argv))
        for k, v in ref.items():
            assert k in mine, k
            assert mine[k] == v, (key, k, mine[k], v)
    a = parsers_parser(str(g["infer_argv"]).split() + ["--trivia", "0"])
    assert a.trivia == 0 and a.num_return_sequences == 100 and a.d_model == 768


@pytest.mark.parametrize("flag,value", [
    ("adaptor_decode", "0"), ("adaptor_efficient", "0"), ("decode_embedding", "1"), ("decode_embedding", "0"),
    ("hierarchic_decode", "1"), ("multiple_decoder", "1"), ("denoising", "1"), ("tie_decode_embedding", "0"),
    ("softmax", "1"), ("gen_method", "top_k"), ("position", "0"), ("model_info", "3b"), ("model_info", "11b")])
def test_reference_model_variants_the_kernels_do_not_implement_fail_loudly(flag, value):
    """main.py forwards these flags into T5Config (main_models.py:748-780) and modeling_t5.py:1578-1640 /
    main_models.py:1350-1397 branch on them; the HIP path implements the shipped setting only, so any other value must stop
    the run before a model is built — through GDRConfig.from_args AND through the `--mode eval` entry point."""
    from gdr_amd import main as gmain
    from gdr_amd.config import GDRConfig, unsupported_variant
    a = gmain.parsers_parser(["--mode", "eval", "--" + flag, value])
    assert flag in unsupported_variant(a)
    with pytest.raises(SystemExit) as e:
        GDRConfig.from_args(a)
    assert "--" + flag in str(e.value)
    with pytest.raises(SystemExit) as e:                     # the entry point stops before touching a GPU or a file
        gmain.main(["--mode", "eval", "--" + flag, value])
    assert "--" + flag in str(e.value)
    # the shipped settings pass
    assert unsupported_variant(gmain.parsers_parser(["--mode", "eval"])) is None


def test_synthetic_corpus_shard_is_bit_identical_to_the_slice_of_the_whole():
    """r06: an N-GPU rank materialises only its own rows of the synthetic corpus (synth.make_corpus(rows=), bench.py / main.py): the
    shard must be the slice of the whole corpus BIT FOR BIT — every chunk geometry (shard inside one chunk, spanning chunks, starting /
    ending on a chunk edge, empty, the whole) — and make_gold must reproduce make_queries' gold ids without the corpus."""
    from gdr_amd import synth
    from gdr_amd.dist import shard_bounds
    N, d, chunk = 5000, 32, 1024
    whole = synth.make_corpus(N, d, chunk=chunk)
    for lo, hi in ((0, N), (0, 700), (100, 900), (1000, 2100), (1024, 2048), (3000, 5000), (4999, 5000), (2500, 2500)):
        part = synth.make_corpus(N, d, chunk=chunk, rows=(lo, hi))
        assert part.shape == (hi - lo, d) and np.array_equal(part, whole[lo:hi]), (lo, hi)
    for world in (2, 3, 8):
        parts = [synth.make_corpus(N, d, chunk=chunk, rows=shard_bounds(N, world, r, cluster_size=12)) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), whole)
    with pytest.raises(ValueError):
        synth.make_corpus(N, d, rows=(10, N + 1))
    _, gold = synth.make_queries(whole, 77)
    assert np.array_equal(gold, synth.make_gold(N, 77))


def _gloo_worker(rank, world, port, tmp):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdr_amd import synth
        from gdr_amd.dist import ShardedIndex, shard_bounds
        from oracle import retrieval_ref
        N, d, k, B_local = 5003, 32, 20, 6                   # 417 clusters of 12 (the last one short): uneven shards
        D = synth.make_corpus(N, d, seed=4)
        Q, _ = synth.make_queries(D, B_local * world, seed=5)
        lo, hi = shard_bounds(N, world, rank, cluster_size=12)
        flagged = (world - 1, 2 * B_local - 1 if world > 1 else 0)   # (shard, global query) whose list "overflowed"

        # CPU stand-ins for the HIP ops with the same contracts (the collective logic under test is the product code)
        def local_topk(Qa, Ds, kk, off):
            v, i = retrieval_ref.sim_topk(Qa, Ds, kk)
            st = torch.zeros(Qa.shape[0], dtype=torch.int32)
            if rank == flagged[0]:
                st[flagged[1]] = 1
            return v, (i + off).to(torch.int32), st

        def pack(v, i, st):
            B, kk = v.shape
            pairs = np.zeros((B, kk + 1, 2), dtype=np.int32)
            pairs[:, :kk, 0] = v.numpy().view(np.int32)
            pairs[:, :kk, 1] = i.numpy()
            pairs[:, kk, 1] = st.numpy()
            return torch.from_numpy(pairs.view(np.int64).reshape(B, kk + 1))

        def merge_packed(pairs):
            G, B, k1 = pairs.shape
            kk = k1 - 1
            raw = pairs.numpy().view(np.int32).reshape(G, B, k1, 2)
            v = np.ascontiguousarray(raw[:, :, :kk, 0]).view(np.float32).transpose(1, 0, 2).reshape(B, G * kk)
            i = raw[:, :, :kk, 1].transpose(1, 0, 2).reshape(B, G * kk)
            key = np.lexsort((i, -v), axis=1)[:, :kk]                     # higher score, then lower id
            st = (raw[:, :, kk, 1] != 0).any(axis=0).astype(np.int32)
            return (torch.from_numpy(np.take_along_axis(v, key, 1)), torch.from_numpy(np.take_along_axis(i, key, 1)),
                    torch.from_numpy(st))

        index = ShardedIndex(torch.from_numpy(D[lo:hi]), lo, local_topk=local_topk, pack=pack, merge_packed=merge_packed)
        q_all = index.gather_queries(torch.from_numpy(Q[rank * B_local:(rank + 1) * B_local]))
        assert torch.equal(q_all, torch.from_numpy(Q))
        v, i, st = index.search(q_all, k, return_status=True)
        rv, ri = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), k)
        vo, io, so = index.search_own(q_all, k, return_status=True)      # all-to-all form: this rank's query block only
        va, ia, sa = index.search_own_async(q_all, k).wait()
        blk = slice(rank * B_local, (rank + 1) * B_local)
        want_st = torch.zeros(B_local * world, dtype=torch.int32)
        want_st[flagged[1]] = 1
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([
            int(torch.equal(i.to(torch.int64), ri)), int(torch.allclose(v, rv, atol=1e-6)),
            int(torch.equal(io, i[blk]) and torch.equal(vo, v[blk])),
            int(torch.equal(ia, io) and torch.equal(va, vo) and torch.equal(sa, so)),
            int(torch.equal(st, want_st) and torch.equal(so, want_st[blk])), hi - lo]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_search_gloo(tmp_path, world):
    """world_size-2 and -4 runs of gdr_amd/dist.py on CPU/gloo with uneven shards: query all-gather, per-shard top-k
    with id offsets, ONE collective of the packed (score, id, status) lists and the merge reproduce the single-shard
    result on every rank — replicated form (all-gather), serving form (all-to-all), its async variant — and a shard's
    overflow flag reaches exactly the rows it concerns."""
    import socket
    import torch.multiprocessing as mp
    from gdr_amd.dist import shard_bounds
    assert shard_bounds(100, 3, 0, 12) == (0, 36) and shard_bounds(100, 3, 2, 12) == (72, 100)
    assert [shard_bounds(320000, 8, r, 12) for r in (0, 7)] == [(0, 40008), (280008, 320000)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"ok{r}.npy").tolist() for r in range(world)]
    assert all(r[:5] == [1, 1, 1, 1, 1] for r in rows), rows
    sizes = [r[5] for r in rows]
    assert sum(sizes) == 5003 and len(set(sizes)) > 1, sizes           # the shards really were uneven


def _cpu_rerank(q, D, offs, ids, beam, alphas, k, lo, hi, func, positions):
    """CPU stand-in of ops.rerank_topk in the per-query block layout (test infrastructure: the collective logic of
    dist.ShardedIndex.rerank_own is what runs as product code).  A candidate's score is computed from its own row alone
    (float64 dot, rounded once), so it cannot depend on the shard — the property the HIP kernel has by construction."""
    import numpy as np
    import torch
    qn, Dn, on, idn, bn = q.numpy(), D.numpy(), offs.numpy(), ids.numpy(), beam.numpy()
    B, R = bn.shape
    A = len(alphas)
    vals = np.full((B, A, k), -np.inf, np.float32)
    out = np.full((B, A, k), -1, np.int32)
    for b in range(B):
        p = torch.softmax(torch.from_numpy(bn[b]), dim=-1).numpy()
        cand = []
        for j in range(R):
            for c in range(on[b, j], on[b, j + 1]):
                doc = int(idn[b, c])
                if lo <= doc < hi:
                    x = np.float32((qn[b].astype(np.float64) * Dn[doc - lo].astype(np.float64)).sum())
                    cand.append((c, j, np.float32(np.tanh(x)) if func == "tanh" else np.float32(1 / (1 + np.exp(-x)))))
        for ai, al in enumerate(alphas):
            scored = sorted(((np.float32(s + np.float32(np.float32(al) * p[j])), c) for c, j, s in cand), key=lambda t: (-t[0], t[1]))
            for r, (sc, c) in enumerate(scored[:k]):
                vals[b, ai, r] = sc
                out[b, ai, r] = c if positions else idn[b, c]
    return torch.from_numpy(vals), torch.from_numpy(out)


def _cpu_wire():
    """CPU stand-ins of ops.rerank_wire_pack / rerank_wire_unpack / rerank_positions_to_ids (test infrastructure)."""
    import torch
    i32 = torch.int32

    def pack(q, beam, offs, ids):
        return torch.cat([q.contiguous().view(i32), beam.contiguous().view(i32), offs.to(i32), ids.to(i32)], 1).contiguous()

    def unpack(rows, d, R, stride):
        return (rows[:, :d].contiguous().view(torch.float32), rows[:, d:d + R].contiguous().view(torch.float32),
                rows[:, d + R:d + 2 * R + 1].contiguous(), rows[:, d + 2 * R + 1:].contiguous())

    def pos2id(pos, ids):
        p = pos.long()
        return torch.where(p >= 0, ids.long().gather(1, p.clamp(min=0)), p).to(i32)

    return pack, unpack, pos2id


def _gloo_rerank_worker(rank, world, port, tmp):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdr_amd import synth
        from gdr_amd.dist import ShardedIndex, shard_bounds
        N, d, R, Bl, k, csz = 5003, 32, 4, 3, 14, 12           # 417 clusters of 12 (the last one short): uneven shards
        stride = R * csz
        D = synth.make_corpus(N, d, seed=4)
        B = Bl * world
        Q, _ = synth.make_queries(D, B, seed=5)
        Q *= 0.2
        rng = np.random.Generator(np.random.PCG64(77))         # the same draw on every rank
        n_cl = (N + csz - 1) // csz
        offs = np.zeros((B, R + 1), np.int32)
        ids = np.full((B, stride), -7, np.int32)               # garbage past a query's list must never be read
        for b in range(B):
            pos = 0
            for j in range(R):
                if b == 1:
                    cl = j                                     # every candidate of this query lives on rank 0
                elif b == 2 and j > 0:
                    cl = -1                                    # three empty segments: fewer than k candidates over all
                elif (b + j) % 5 == 0:
                    cl = -1
                else:
                    cl = int(rng.integers(0, n_cl))
                if cl >= 0:
                    mem = np.arange(cl * csz, min((cl + 1) * csz, N), dtype=np.int32)
                    ids[b, pos:pos + mem.size] = mem
                    pos += mem.size
                offs[b, j + 1] = pos
        beam = rng.standard_normal((B, R)).astype(np.float32)
        alphas = [0, 0.5, 1, 3]
        lo, hi = shard_bounds(N, world, rank, cluster_size=csz)

        def pack(v, i, st):
            Bq, kk = v.shape
            pairs = np.zeros((Bq, kk + 1, 2), dtype=np.int32)
            pairs[:, :kk, 0] = v.numpy().view(np.int32)
            pairs[:, :kk, 1] = i.numpy()
            return torch.from_numpy(pairs.view(np.int64).reshape(Bq, kk + 1))

        def merge_packed(pairs):
            G, Bq, k1 = pairs.shape
            kk = k1 - 1
            raw = pairs.numpy().view(np.int32).reshape(G, Bq, k1, 2)
            v = np.ascontiguousarray(raw[:, :, :kk, 0]).view(np.float32).transpose(1, 0, 2).reshape(Bq, G * kk)
            i = raw[:, :, :kk, 1].transpose(1, 0, 2).reshape(Bq, G * kk)
            big = np.where(i < 0, np.iinfo(np.int32).max, i)                # padding entries (-inf, -1) sort last
            key = np.lexsort((big, -v), axis=1)[:, :kk]                     # higher score, then lower position
            return (torch.from_numpy(np.take_along_axis(v, key, 1)), torch.from_numpy(np.take_along_axis(i, key, 1)),
                    torch.zeros(Bq, dtype=torch.int32))

        index = ShardedIndex(torch.from_numpy(D[lo:hi]), lo, local_topk=lambda *a: None, pack=pack, merge_packed=merge_packed,
                             local_rerank=_cpu_rerank, wire=_cpu_wire())
        blk = slice(rank * Bl, (rank + 1) * Bl)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))          # noqa: E731
        v, i = index.rerank_own(T(Q[blk]), T(offs[blk]), T(ids[blk]), T(beam[blk]), alphas, k)
        rv, ri = _cpu_rerank(T(Q[blk]), T(D), T(offs[blk]), T(ids[blk]), T(beam[blk]), alphas, k, 0, N, "tanh", False)
        short = int((ri[2 - rank * Bl] < 0).any()) if rank == 0 else 1     # query 2 has fewer than k candidates
        np.save(os.path.join(tmp, f"rr{rank}.npy"), np.array([int(torch.equal(v, rv)), int(torch.equal(i, ri)), short, hi - lo,
                                                              int((ri >= 0).sum())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_gdr_rerank_gloo(tmp_path, world):
    """GDR mode of gdr_amd/dist.py (SURVEY §8e second half; main_models.py:1434-1462,1574-1637 sharded) on CPU/gloo with
    uneven, cluster-aligned shards: ONE all-gather of queries + candidate blocks, per-shard {score, position} lists, ONE
    all-to-all, merge — bit-identical to the unsharded rerank on every rank, including a query whose candidates all live
    on one rank, empty segments and a query with fewer than k candidates."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_rerank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"rr{r}.npy").tolist() for r in range(world)]
    assert all(r[:3] == [1, 1, 1] for r in rows), rows
    assert len({r[3] for r in rows}) > 1 and all(r[4] > 0 for r in rows), rows


class _CpuDci:
    """CPU stand-in of ops.DeviceClusterIndex.candidates (the gdr_cluster_candidates kernel): the host string path."""

    def __init__(self, index, args, max_cluster):
        self.index, self.args, self.max_cluster = index, args, max_cluster

    def candidates(self, out_ids, B, R):
        import torch
        from gdr_amd import codec
        stride = R * self.max_cluster
        dec = codec.dec_2d(codec.decode_token(self.args, out_ids.numpy()), R)
        offs = np.zeros((B, R + 1), np.int32)
        ids = np.full((B, stride), -7, np.int32)
        for b in range(B):
            pos = 0
            for j in range(R):
                mem = np.asarray(self.index[dec[b][j]], np.int32)
                ids[b, pos:pos + mem.size] = mem
                pos += mem.size
                offs[b, j + 1] = pos
        return None, torch.from_numpy(offs), torch.from_numpy(ids), stride


def _gloo_retriever_worker(rank, world, port, tmp):
    import os
    import types
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdr_amd import codec, synth
        from gdr_amd.dist import ShardedIndex, shard_bounds
        from gdr_amd.modeling import GDRRetriever
        V, depth, csz, d, R, Bl, L, ml = 5, 3, 7, 32, 4, 3, 6, 6
        n_cl = V ** depth
        N = n_cl * csz - 3                                   # the last cluster is short
        names = ["-".join(str(x) for x in synth.cluster_digits(c, depth, V)) for c in range(n_cl)]
        offsets = np.minimum(np.arange(n_cl + 1) * csz, N).astype(np.int32)
        index = codec.ClusterIndex(names, offsets, np.arange(N, dtype=np.int32))
        D = synth.make_corpus(N, d, cluster_size=csz, seed=4)
        args = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=V, max_output_length=ml, length_penalty=0.8,
                                     kary=V, position=1, score_rate=[0, 0.5, 2], loss_func="tanh")
        rng = np.random.Generator(np.random.PCG64(1000 + rank))    # every rank decodes ITS OWN queries

        class FakeModel:                                          # stand-in of GDRModel._generate_launch (encoder + beam decode)
            device = torch.device("cpu")

            def _generate_launch(self, input_ids, mask, num_beams, max_length, length_penalty, nret):
                B = input_ids.shape[0]
                enc_h = torch.from_numpy(rng.standard_normal((B, L, d)).astype(np.float32) * 0.3)
                ids = np.zeros((B * nret, max_length), np.int64)
                for r in range(B * nret):
                    if r % 5 == 4:                                # a row that names no cluster: depth too short
                        toks = codec.encode_single_newid(args, "1-2")
                    else:
                        toks = codec.encode_single_newid(args, names[int(rng.integers(0, n_cl))])
                    ids[r, 1:1 + len(toks)] = toks
                lens = torch.full((B * nret,), depth + 1, dtype=torch.int32)
                scores = torch.from_numpy(-np.sort(rng.random(B * nret))).double()
                return enc_h, torch.from_numpy(ids), lens, scores

        lo, hi = shard_bounds(N, world, rank, cluster_size=csz)
        pack, merge_packed = _np_pack_merge()
        sharded = ShardedIndex(torch.from_numpy(D[lo:hi]), lo, local_topk=lambda *a: None, pack=pack, merge_packed=merge_packed,
                               local_rerank=_cpu_rerank, wire=_cpu_wire())
        retr = GDRRetriever(FakeModel(), None, index, args, sharded=sharded)
        retr._dci = _CpuDci(index, args, csz)
        ok, n_docs = 1, 0
        for step in range(2):                                     # two steps: the collectives stay in lockstep
            batch = {"source_ids": torch.zeros((Bl, L), dtype=torch.int64), "texts": [f"q{rank}_{step}_{b}" for b in range(Bl)]}
            state = retr._step_launch(batch)
            out = retr._step_finish(state)
            # expectation: the unsharded rerank over the whole corpus for this rank's own queries
            _c, offs, ids, stride = retr._dci.candidates(state["ids"], Bl, R)
            q = state["enc_h"][:, 0].contiguous()
            beam = state["scores"].to(torch.float32).view(Bl, R)
            rv, ri = _cpu_rerank(q, torch.from_numpy(D), offs, ids, beam, args.score_rate, R, 0, N, "tanh", False)
            ok &= int(torch.equal(out["rerank_values"], rv))
            want = [[[str(int(x)) for x in ri[b, ai]] for ai in range(3)] for b in range(Bl)]
            ok &= int(out["doc_ids"] == want)
            ok &= int(len(out["inf_index_batch"]) == Bl and out["inf_index_batch"][0][1][0][1] == ",".join(want[0][1]))
            n_docs += int((ri >= 0).sum())
        np.save(os.path.join(tmp, f"gr{rank}.npy"), np.array([ok, n_docs, hi - lo]))
    finally:
        dist.destroy_process_group()


def _np_pack_merge():
    """numpy stand-ins of ops.topk_pack / ops.topk_merge_packed (the wire form of gdr_hip.h; padding entries sort last)."""
    import torch

    def pack(v, i, st):
        Bq, kk = v.shape
        pairs = np.zeros((Bq, kk + 1, 2), dtype=np.int32)
        pairs[:, :kk, 0] = v.numpy().view(np.int32)
        pairs[:, :kk, 1] = i.numpy()
        return torch.from_numpy(pairs.view(np.int64).reshape(Bq, kk + 1))

    def merge_packed(pairs):
        G, Bq, k1 = pairs.shape
        kk = k1 - 1
        raw = pairs.numpy().view(np.int32).reshape(G, Bq, k1, 2)
        v = np.ascontiguousarray(raw[:, :, :kk, 0]).view(np.float32).transpose(1, 0, 2).reshape(Bq, G * kk)
        i = raw[:, :, :kk, 1].transpose(1, 0, 2).reshape(Bq, G * kk)
        big = np.where(i < 0, np.iinfo(np.int32).max, i)
        key = np.lexsort((big, -v), axis=1)[:, :kk]
        return (torch.from_numpy(np.take_along_axis(v, key, 1)), torch.from_numpy(np.take_along_axis(i, key, 1)),
                torch.zeros(Bq, dtype=torch.int32))

    return pack, merge_packed


@pytest.mark.parametrize("world", [2])
def test_sharded_two_stage_retriever_gloo(tmp_path, world):
    """The COMPOSED sharded two-stage path (BASELINE config C5's layout; main_models.py:1337-1642 with :1434-1462,1574-1637
    sharded) under gloo at world size 2: GDRRetriever(sharded=ShardedIndex) — every rank decodes its own queries, builds
    their candidate blocks, rerank_own exchanges / scores / merges, positions map back to doc ids, the host formats the
    reference's step output.  Stand-ins for the compute (encoder + beam decode, cluster lookup, per-shard rerank); the
    retriever and the collective logic are the product code.  Equals the unsharded rerank over the whole corpus bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gloo_retriever_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"gr{r}.npy").tolist() for r in range(world)]
    assert all(r[0] == 1 and r[1] > 0 for r in rows), rows


def test_bench_refuses_rccl_with_more_ranks_than_gpus():
    """`--backend nccl` (RCCL, the product backend) with WORLD_SIZE above the node's GPU count must fail loudly instead of folding
    ranks onto the GPUs that exist: this container has no GPU, so a 2-rank nccl environment is exactly that case.  The refusal
    happens before anything touches a device."""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "nccl", "--corpus", "3000"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert out.returncode != 0 and "needs 2 GPUs" in out.stderr, (out.returncode, out.stderr[-500:])
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], "no JSON line from a refused run"


def test_launcher_timeout_kills_the_whole_rank_group(tmp_path):
    """A hung launch is bounded by `timeout=` and leaves nobody behind: the launcher and its ranks run in a process group of their
    own, the watchdog terminates the launcher and then SIGKILLs the group (a SIGKILL to torch.distributed.run alone would orphan
    the ranks, which keep the GPUs and the stdout pipe: spawn_ranks would never return)."""
    import time
    from gdr_amd import launch
    stub = tmp_path / "hang_stub.py"
    stub.write_text(
        "import os, signal, sys, time\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"          # a rank that does not even honour SIGTERM
        "open(os.path.join(sys.argv[1], 'pid%s' % os.environ['RANK']), 'w').write(str(os.getpid()))\n"
        "print('[rank] up', flush=True)\n"
        "time.sleep(600)\n")
    t0 = time.monotonic()
    rc, text = launch.spawn_ranks(2, [str(tmp_path)], script=str(stub), relay=False, timeout=6)
    took = time.monotonic() - t0
    assert rc != 0 and took < 60, (rc, took)
    pids = [int(f.read_text()) for f in tmp_path.glob("pid*")]
    assert len(pids) == 2
    time.sleep(0.5)
    for pid in pids:
        alive = os.path.exists(f"/proc/{pid}") and "Z" not in open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0]
        assert not alive, f"rank process {pid} survived the launcher's timeout"


def test_launcher_spawns_fresh_ranks_and_relays_one_json_line(tmp_path, monkeypatch, capsys):
    """`python bench.py --gpus N` with no RANK in the environment (the driver's multi-GPU command; the reference's analogue of
    one command fanning out over the GPUs is Data_process/NQ_dataset/bert/bert_NQ.sh:5-12): gdr_amd.launch.spawn_ranks starts
    N fresh rank processes under torch.distributed.run — proven here with a stub rank script on CPU: two children with RANK
    0 / 1 and WORLD_SIZE 2 ran, one JSON line came back, a failing rank makes the launcher's code non-zero — and bench.main()
    routes `--gpus 2` into it with its own path as the script, lets only the JSON line through to stdout and exits with the
    launcher's code."""
    import json
    from gdr_amd import launch
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "open(os.path.join(sys.argv[1], f'seen{rank}'), 'w').write(os.environ['MASTER_ADDR'] + ' ' + ' '.join(sys.argv[2:]))\n"
        "print(f'[rank {rank}] noise')\n"
        "if '--fail' in sys.argv and rank == 1: sys.exit(3)\n"
        "if rank == 0: print(json.dumps({'metric': 'stub', 'n_gpus': world}))\n")
    rc, text = launch.spawn_ranks(2, [str(tmp_path), "--steps", "2"], script=str(stub), relay=False)
    assert rc == 0, text
    assert sorted(f.name for f in tmp_path.glob("seen*")) == ["seen0", "seen1"]
    assert (tmp_path / "seen1").read_text() == "127.0.0.1 --steps 2"
    js = [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]
    assert js == [{"metric": "stub", "n_gpus": 2}]
    rc, _ = launch.spawn_ranks(2, [str(tmp_path), "--fail"], script=str(stub), relay=False)
    assert rc != 0
    got = []
    rc, _ = launch.spawn_ranks(2, [str(tmp_path)], script=str(stub), relay=got.append)      # a callable gets every line
    assert rc == 0 and sum(ln.startswith("{") for ln in got) == 1 and sum("noise" in ln for ln in got) == 2
    assert capsys.readouterr().out == "", "a relay callable decides where lines go: nothing is echoed behind its back"
    # bench.py's side of it
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_spawn(n, argv, script=None, module=None, relay=True, **kw):
        seen.update(n=n, argv=list(argv), script=script)
        relay("[rank 1] noise\n")
        relay('{"metric": "x"}\n')
        return 5, ""

    monkeypatch.setattr(launch, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--launcher"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5
    assert seen["n"] == 2 and seen["argv"] == ["--gpus", "2", "--steps", "3"] and seen["script"] == os.path.join(REPO, "bench.py")
    cap = capsys.readouterr()
    assert cap.out == '{"metric": "x"}\n' and "noise" in cap.err


@pytest.mark.parametrize("n,bs,world", [(10, 4, 1), (10, 4, 2), (10, 4, 3), (10, 4, 4), (64, 8, 8), (3, 4, 4), (9, 3, 2)])
def test_n_gpu_batch_dealing_round_trips(n, bs, world):
    """`main.py --n_gpu N` (the reference's analogue: Lightning's DistributedSampler over the validation set,
    main_models.py:1990-1999): every rank must run the same number of fixed-size steps for the sharded stage 2's collectives;
    the real steps, put back in span order on rank 0, must cover every query exactly once, in order."""
    from gdr_amd.main import deal_spans, reassemble_spans
    per_rank, steps = [], set()
    for rank in range(world):
        all_spans, mine, real = deal_spans(n, bs, world, rank)
        steps.add(len(mine))
        assert len(mine) == len(real) and all(0 <= lo < hi <= n and hi - lo <= bs for lo, hi in mine)
        per_rank.append([list(range(lo, hi)) for (lo, hi), r in zip(mine, real) if r])    # a step's "output": its query ids
    assert len(steps) == 1, "every rank runs the same number of steps"
    back = reassemble_spans(per_rank, len(all_spans), world)
    assert [q for span in back for q in span] == list(range(n))
    assert sum(len(p) for p in per_rank) == len(all_spans)


def test_trie_flattening_matches_reference_treebuilder_semantics():
    """codec.Trie (flat arrays for the device) encodes exactly the nested children of TreeBuilder.add
    (main_models.py:135-151) for the docid sequences of the reference-made fixture."""
    from gdr_amd import codec
    from oracle import beam_ref
    g = golden("g11_beam_trie")
    for case in "abc":
        V = int(g[f"{case}_meta"][0])
        seqs = g[f"{case}_seqs"].tolist()
        t = codec.Trie.from_sequences(seqs, V)
        nested = beam_ref.build_trie(seqs)

        def walk(node_dict, node, depth):
            kids = {tok for tok in node_dict}
            flat = {depth * V + 2 + c for c in range(V) if t.child[node, c] >= 0} | ({1} if t.eos_ok[node] else set())
            assert kids == flat, (case, depth, kids, flat)
            for tok, sub in node_dict.items():
                if tok != 1:
                    walk(sub, int(t.child[node, tok - (depth * V + 2)]), depth + 1)

        walk(nested, 0, 0)


def test_trie_breadth_first_relabelling_keeps_the_language():
    """codec.Trie.breadth_first (the node order of the device prefix table, gdr_hip.h GdrPrefixTable): same accepted docids,
    every depth a contiguous id range, parent / token arrays consistent with the child table."""
    from gdr_amd import codec, synth
    V = 6
    names, depth, _, _ = synth.make_cluster_ids(5000, cluster_size=12, V=V)      # 417 ids of depth 4
    names = [n for i, n in enumerate(names) if i % 3]                              # holes
    t = codec.Trie.from_docids(names, V)
    b, level_off, parent, tok = t.breadth_first()
    assert b.child.shape == t.child.shape and int(level_off[0]) == 0 and int(level_off[1]) == 1 and parent[0] == -1
    assert int(level_off[-1]) == b.child.shape[0] and len(level_off) == depth + 2
    for s in range(len(level_off) - 1):                                            # children of level s live in level s+1
        kids = b.child[level_off[s]:level_off[s + 1]]
        kids = kids[kids >= 0]
        if s + 2 < len(level_off):
            assert kids.min() >= level_off[s + 1] and kids.max() < level_off[s + 2]
        else:
            assert kids.size == 0
    for node in range(1, b.child.shape[0]):
        p = int(parent[node])
        d = int(np.searchsorted(level_off, p, side="right") - 1)                   # depth of the parent
        c = int(tok[node]) - (d * V + 2)
        assert 0 <= c < V and b.child[p, c] == node

    def accepts(trie, name):
        n = 0
        for d, c in enumerate(int(x) for x in name.split("-")):
            n = trie.child[n, c]
            if n < 0:
                return False
        return bool(trie.eos_ok[n])

    probe = names[:40] + ["0-0-0-5", "5-5-5-5", "1-2"]
    assert [accepts(t, n) for n in probe] == [accepts(b, n) for n in probe] and all(accepts(b, n) for n in names[:40])


def test_main_calculate_mode_recomputes_the_metrics_of_a_tsv(tmp_path, capsys):
    """`--mode calculate` (main.py:253-258, 495-496): recall / MRR100 of an existing res1 TSV, no GPU involved."""
    from gdr_amd import main as gmain
    g = golden("g7_metrics")
    path = tmp_path / "res1.tsv"
    with open(path, "w") as f:
        for r in g["rows"]:
            f.write("\t".join(str(x) for x in r) + "\n")
    rec, mrr = gmain.main(["--mode", "calculate", "--res1_save_path", str(path), "--recall_num", "1", "5", "10", "20", "50", "100"])
    assert rec == pytest.approx(float(g["recall_v"][-1])) and mrr == pytest.approx(float(g["mrr100"]))
    out = capsys.readouterr().out
    assert "recall@1:" in out and "MRR100:" in out


def test_artifact_converters_on_synthetic_pickles(tmp_path):
    """tools/convert_artifacts.py on pickles shaped like the reference's (doc_embedding.pkl: list of [1,d] tensors;
    indexmap.pkl: dict cluster-string -> doc ids; Lightning ckpt with model./encoder.model. prefixes)."""
    import importlib.util
    import pickle
    import torch
    spec = importlib.util.spec_from_file_location("convert_artifacts", os.path.join(REPO, "tools", "convert_artifacts.py"))
    ca = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ca)
    g = np.random.Generator(np.random.PCG64(1))
    emb = [torch.from_numpy(g.standard_normal((1, 16)).astype(np.float32)) for _ in range(7)]
    arr = ca.convert_doc_embedding(pickle.loads(pickle.dumps(emb)))
    assert arr.shape == (7, 16) and arr.dtype == np.float32 and np.array_equal(arr[3], emb[3].numpy()[0])
    assert np.array_equal(ca.convert_doc_embedding({i: e[0] for i, e in enumerate(emb)}), arr)
    im = {"3-1": [5, 2], "0-0": [1], "29-4-7": [0, 3, 4, 6]}
    z = ca.convert_indexmap(pickle.loads(pickle.dumps(im)))
    from gdr_amd import codec
    idx = codec.ClusterIndex([str(x) for x in z["cluster_names"]], z["cluster_offsets"], z["cluster_members"])
    assert all(idx[k] == v for k, v in im.items())
    ckpt = {"state_dict": {"model.shared.weight": torch.zeros(2, 2), "model.encoder.block.0.x": torch.ones(1),
                           "encoder.model.ctx_encoder.bert_model.embeddings.word_embeddings.weight": torch.ones(3)}}
    t5, tower = ca.split_checkpoint(ckpt)
    assert set(t5) == {"shared.weight", "encoder.block.0.x"}
    assert set(tower) == {"ctx_encoder.bert_model.embeddings.word_embeddings.weight"}
