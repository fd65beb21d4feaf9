"""Row-sharded corpus search across the GPUs of one node (BASELINE config C4/C5).

One process per GPU.  Rank r holds the contiguous row block D[lo:hi] (whole clusters when a cluster size is
given, so a cluster never straddles ranks).  A search step is
    all-gather of the per-rank query embeddings  ->  local fused sim+top-k on the shard (doc ids offset by lo)
    -> pack every query's list into its wire form  int64[B, k+1] = k x {fp32 score, int32 id} + {0, status}
    -> ONE collective of that buffer (RCCL over xGMI)
    -> local merge [G,B,k+1] -> (values, ids, status)[B,k], bit-identical on every rank (tie rule: higher score, then
       lower id; status = OR of the shards' overflow flags).
`search_own` is the serving form: a rank only needs the answers to its OWN queries, so the exchange is one all-to-all
(rank j receives, from every rank, the lists of j's query block: B_local*(k+1)*8 bytes per peer instead of B*(k+1)*8)
and each rank merges [G,B_local,k+1] — 1/G of the traffic and of the merge work of the replicated form `search`
(one all-gather, every rank merges all queries).  `search_own_async` issues pack + exchange + merge on a side stream
and returns a handle: the caller's stream is free to run the next batch's encoder meanwhile and joins with
`handle.wait()`.
The reference has no inference-time collective; its closest analogues are the training-time all_gather of
reps (GDR_model/encoder.py:134-145) and the offline per-GPU partitioning of Data_process/NQ_dataset/bert/bert.py:51-61.

Overflow contract (gdr_hip.h, gdr_sim_topk): with exact=True (default) the local search repairs an overflowed query
exhaustively before the exchange (one status read-back per step), so every status is 0; with exact=False nothing
synchronises and the merged status tells the caller which rows are the top-k of a subset.

GDR mode (two-stage retrieval, BASELINE config C5; the arithmetic of main_models.py:1434-1462,1574-1637) shards the same
way — whole clusters per rank, so a cluster never straddles ranks: `rerank_own` all-gathers every rank's queries with their
decoded candidate lists (ONE fixed-size buffer), each rank scores the candidates whose doc ids fall in its [lo, hi) and
keeps a per-(query, alpha) top-k of {score, candidate position}, ONE all-to-all hands every rank the per-shard lists of its
own queries, and the merge by "higher score, then lower position" equals the unsharded rerank bit for bit (a candidate's
score does not depend on the shard that computed it).

The compute callables default to the HIP ops; tests inject CPU stand-ins to exercise the collective logic
under gloo (there is no CPU compute path in the product).
"""
import torch
import torch.distributed as dist


def shard_bounds(N, world, rank, cluster_size=1):
    """Contiguous [lo, hi) of rank `rank`; boundaries fall on multiples of cluster_size."""
    units = (N + cluster_size - 1) // cluster_size
    per, rem = divmod(units, world)
    lo_u = rank * per + min(rank, rem)
    hi_u = lo_u + per + (1 if rank < rem else 0)
    return min(lo_u * cluster_size, N), min(hi_u * cluster_size, N)


class _Pending:
    """Result of search_own_async: tensors produced on a side stream; wait() joins the caller's current stream."""

    def __init__(self, result, stream):
        self._result, self._stream = result, stream

    def wait(self):
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            for t in self._result:
                t.record_stream(torch.cuda.current_stream())
        return self._result


class ShardedIndex:
    def __init__(self, D_shard, lo, group=None, local_topk=None, pack=None, merge_packed=None, exact=True, local_rerank=None,
                 wire=None):
        """local_topk(Q, D, k, idx_offset) -> (values [B,k], ids int32 [B,k], status int32 [B]);
        pack(values, ids, status) -> int64 [B,k+1];  merge_packed(pairs [G,B,k+1]) -> (values, ids, status);
        local_rerank(q, D, cand_offsets [B,R+1], cand_ids [B,stride], beam_scores, alphas, k, lo, hi, func, positions)
        -> (values [B,A,k], int32 [B,A,k]) — ops.rerank_topk in the per-query block layout;
        wire = (wire_pack(q, beam, offs, ids) -> int32 [B,W], wire_unpack(rows, d, R, stride) -> (q, beam, offs, ids),
        positions_to_ids(pos [B,...], ids [B,stride]) -> int32 ids) — ops.rerank_wire_pack / _unpack / rerank_positions_to_ids."""
        # D: what local_topk searches (a tensor, or an ops.PrefilteredCorpus around it); rows: the raw tensor for everything
        # that is not a search (the rerank gathers, shapes)
        self.D, self.lo, self.group = D_shard, int(lo), group
        self.rows = getattr(D_shard, "D", D_shard) if not isinstance(D_shard, torch.Tensor) else D_shard
        if wire is None:
            from . import ops as _wops
            wire = (_wops.rerank_wire_pack, _wops.rerank_wire_unpack, _wops.rerank_positions_to_ids)
        self.wire_pack, self.wire_unpack, self.positions_to_ids = wire
        if local_topk is None or pack is None or merge_packed is None:
            from . import ops
            ws = ops.Workspace(D_shard.device)
            local_topk = local_topk or (lambda Q, D, k, off: ops.sim_topk(
                Q, D, k, idx_offset=off, workspace=ws, return_status=True, exact_on_overflow=exact))
            pack = pack or ops.topk_pack
            merge_packed = merge_packed or (lambda pairs: ops.topk_merge_packed(pairs, return_status=True))
        if local_rerank is None:
            from . import ops as _ops
            local_rerank = (lambda q, D, offs, ids, beam, alphas, k, lo, hi, func, positions: _ops.rerank_topk(
                q, D, offs, ids, beam, alphas, k, func=func, max_cand=_ops.block_max_cand(offs, beam.shape[1], ids.shape[1]),
                doc_range=(lo, hi), positions=positions, cand_stride=ids.shape[1]))
        self.local_topk, self.pack, self.merge_packed, self.local_rerank = local_topk, pack, merge_packed, local_rerank
        self.distributed = dist.is_initialized()      # a 1-rank group still runs the collectives (exercises RCCL)
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self._side = None

    def gather_queries(self, q_local):
        """[B_local,d] per rank -> [world*B_local,d], rank-major (every rank contributes the same count)."""
        if not self.distributed:
            return q_local
        out = torch.empty((self.world * q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype,
                          device=q_local.device)
        dist.all_gather_into_tensor(out, q_local.contiguous(), group=self.group)
        return out

    @staticmethod
    def _ret(v, i, st, return_status):
        return (v, i, st) if return_status else (v, i)

    def search(self, q_all, k, return_status=False):
        """q_all [B,d] identical on every rank -> (values [B,k], global doc ids int32 [B,k][, status]), identical on
        every rank.  ONE all-gather of the packed per-shard lists."""
        v, i, st = self.local_topk(q_all, self.D, k, self.lo)
        if not self.distributed:
            return self._ret(v, i, st, return_status)
        B = q_all.shape[0]
        mine = self.pack(v, i, st)                                                  # [B, k+1] int64
        allp = torch.empty((self.world * B, k + 1), dtype=mine.dtype, device=mine.device)   # rank-major concatenation
        dist.all_gather_into_tensor(allp, mine, group=self.group)
        return self._ret(*self.merge_packed(allp.view(self.world, B, k + 1)), return_status)

    def _exchange_own(self, v, i, st, B, k):
        bl = B // self.world
        mine = self.pack(v, i, st)                      # [world*bl, k+1]: block g = my list for rank g's queries
        recv = torch.empty_like(mine)                   # block g = rank g's list for MY queries
        dist.all_to_all_single(recv, mine, group=self.group)
        return self.merge_packed(recv.view(self.world, bl, k + 1))

    def search_own(self, q_all, k, return_status=False):
        """q_all [B,d] identical on every rank (rank-major blocks of B/world queries) -> top-k of THIS rank's query block:
        (values [B/world,k], global doc ids int32 [B/world,k][, status]).  Same lists as search()[rank block].
        ONE all-to-all of the packed per-shard lists."""
        v, i, st = self.local_topk(q_all, self.D, k, self.lo)
        if not self.distributed:
            return self._ret(v, i, st, return_status)
        B = q_all.shape[0]
        if B % self.world:
            raise ValueError("search_own: the gathered batch must hold the same number of queries per rank")
        return self._ret(*self._exchange_own(v, i, st, B, k), return_status)

    def search_own_async(self, q_all, k):
        """search_own with pack + all-to-all + merge issued on a side stream: returns a handle whose wait() yields
        (values, ids, status) and makes the caller's current stream wait for them.  Between the call and wait() the
        caller's stream is free — bench.py runs the next batch's encoder there, hiding the exchange."""
        v, i, st = self.local_topk(q_all, self.D, k, self.lo)
        if not self.distributed:
            return _Pending((v, i, st), None)
        B = q_all.shape[0]
        if B % self.world:
            raise ValueError("search_own_async: the gathered batch must hold the same number of queries per rank")
        if not v.is_cuda:                                # CPU stand-ins (gloo tests): nothing to overlap with
            return _Pending(self._exchange_own(v, i, st, B, k), None)
        if self._side is None:
            self._side = torch.cuda.Stream(device=v.device)
        side, cur = self._side, torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for t in (v, i, st):
                t.record_stream(side)
            out = self._exchange_own(v, i, st, B, k)
        return _Pending(out, side)

    # ------------------------------------------------------------------ GDR mode: sharded in-cluster rerank
    def rerank_own(self, q_local, cand_offsets, cand_ids, beam_scores, alphas, k, func="tanh"):
        """Stage 2 of GDR over the row-sharded corpus for THIS rank's queries.
        q_local fp32 [Bl,d]; cand_offsets int32 [Bl,R+1] / cand_ids int32 [Bl,stride]: the per-query block layout
        (ops.DeviceClusterIndex.candidates — the same stride on every rank: num_beams x largest cluster); beam_scores fp32
        [Bl,R].  Returns (values fp32 [Bl,A,k], doc ids int32 [Bl,A,k]) — bit-identical to the unsharded
        ops.rerank_topk over the whole corpus.  Collectives: ONE all-gather (queries + candidate lists, fixed size), ONE
        all-to-all (per-shard {score, position} lists of B*A rows in the wire form of the brute-force search)."""
        Bl, d = q_local.shape
        R, stride, hi = beam_scores.shape[1], cand_ids.shape[1], self.lo + self.rows.shape[0]
        A = len(alphas)
        if not self.distributed:
            return self.local_rerank(q_local, self.rows, cand_offsets, cand_ids, beam_scores, alphas, k, self.lo, hi, func, False)
        mine = self.wire_pack(q_local, beam_scores, cand_offsets.view(Bl, R + 1), cand_ids.view(Bl, stride))   # [Bl, W] int32
        allb = torch.empty((self.world * Bl, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(allb, mine, group=self.group)
        B = self.world * Bl
        q_all, beam_all, offs_all, ids_all = self.wire_unpack(allb, d, R, stride)
        v, pos = self.local_rerank(q_all, self.rows, offs_all, ids_all, beam_all, alphas, k, self.lo, hi, func, True)
        send = self.pack(v.reshape(B * A, k), pos.reshape(B * A, k), None)          # [B*A, k+1]; block g = rank g's queries
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        mv, mp, _st = self.merge_packed(recv.view(self.world, Bl * A, k + 1))
        ids = self.positions_to_ids(mp.view(Bl, A * k), cand_ids.view(Bl, stride))  # position in MY query's block -> doc id
        return mv.view(Bl, A, k), ids.view(Bl, A, k)
