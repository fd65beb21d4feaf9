"""The N > 1 code paths of gdr_amd/dist.py with REAL compute: two (and four) rank processes share this box's one GPU over the gloo
backend (RCCL refuses several ranks on one device; gloo stages the same collectives through the host), every rank runs the HIP
kernels on its own corpus shard, and the merged results must equal the unsharded kernels' bit for bit.  What the CPU/gloo tests
of tests/test_host_logic.py check with stand-in compute, and what test_gpu_entry.py checks through the entry point, here at
the level of ShardedIndex itself (the reference has no inference-time collective: encoder.py:134-145 is its
closest analogue)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdr_amd import ops, synth
        from gdr_amd.dist import ShardedIndex, shard_bounds
        torch.set_grad_enabled(False)
        dev = torch.device("cuda:0")
        N, d, k, Bl, R, csz = 50003, 768, 100, 24, 10, 12
        D = synth.make_corpus(N, d, seed=4)
        B = Bl * world
        Q, _ = synth.make_queries(D, B, seed=5)
        lo, hi = shard_bounds(N, world, rank, cluster_size=csz)
        D_all = torch.from_numpy(D).to(dev)
        index = ShardedIndex(D_all[lo:hi].contiguous(), lo)
        # ---- brute-force mode: query all-gather, per-shard fused sim + top-k, ONE collective of the packed lists, merge
        q_local = torch.from_numpy(Q[rank * Bl:(rank + 1) * Bl]).to(dev)
        q_all = index.gather_queries(q_local)
        assert torch.equal(q_all.cpu(), torch.from_numpy(Q))
        rv, ri = ops.sim_topk(q_all, D_all, k)                                   # the unsharded kernels on the same GPU
        v, i, st = index.search(q_all, k, return_status=True)
        vo, io, so = index.search_own(q_all, k, return_status=True)
        va, ia, sa = index.search_own_async(q_all, k).wait()
        torch.cuda.synchronize()
        blk = slice(rank * Bl, (rank + 1) * Bl)
        ok = [torch.equal(v, rv) and torch.equal(i, ri), torch.equal(vo, rv[blk]) and torch.equal(io, ri[blk]),
              torch.equal(va, vo) and torch.equal(ia, io), int(st.sum()) + int(so.sum()) + int(sa.sum()) == 0]
        # ---- GDR mode: every rank's own queries + candidate blocks -> rerank_own over the two shards == unsharded rerank
        rng = np.random.Generator(np.random.PCG64(100 + rank))
        n_cl = N // csz
        offs = np.zeros((Bl, R + 1), np.int32)
        ids = np.full((Bl, R * csz), -7, np.int32)
        for b in range(Bl):
            for j in range(R):
                cl = int(rng.integers(0, n_cl))
                ids[b, offs[b, j]:offs[b, j] + csz] = np.arange(cl * csz, (cl + 1) * csz)
                offs[b, j + 1] = offs[b, j] + csz
        beam = torch.from_numpy(np.sort(rng.standard_normal((Bl, R)).astype(np.float32), axis=1)[:, ::-1].copy()).to(dev)
        offs_d, ids_d = torch.from_numpy(offs).to(dev), torch.from_numpy(ids).to(dev)
        alphas = [0, 0.5, 1, 1.5, 2, 2.5, 3]
        ql = (q_local * 0.3).contiguous()
        uv, ui = ops.rerank_topk(ql, D_all, offs_d, ids_d, beam, alphas, R, max_cand=R * csz, cand_stride=R * csz)
        sv, si = index.rerank_own(ql, offs_d, ids_d, beam, alphas, R)
        torch.cuda.synchronize()
        ok += [torch.equal(sv, uv), torch.equal(si, ui)]
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([int(x) for x in ok] + [hi - lo]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_index_with_real_kernels_on_ranks_sharing_one_gpu(tmp_path, world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"ok{r}.npy").tolist() for r in range(world)]
    assert all(r[:6] == [1] * 6 for r in rows), rows
    assert sum(r[6] for r in rows) == 50003
