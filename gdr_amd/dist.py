"""Row-sharded corpus search across the GPUs of one node (BASELINE config C4/C5).

One process per GPU.  Rank r holds the contiguous row block D[lo:hi] (whole clusters when a cluster size is
given, so a cluster never straddles ranks).  A search step is
    all-gather of the per-rank query embeddings  ->  local fused sim+top-k on the shard (doc ids offset by lo)
    -> ONE all-gather of the per-shard (score fp32, id int32)[B,k] lists (RCCL over xGMI; B*k*8 bytes per rank)
    -> local merge [G,B,k] -> [B,k], bit-identical on every rank (tie rule: higher score, then lower id).
`search_own` is the serving form: a rank only needs the answers to its OWN queries, so the per-shard lists are exchanged
with one all-to-all (rank j receives, from every rank, the lists of j's query block: B_local*k*8 bytes per peer instead
of B*k*8) and each rank merges [G,B_local,k] — 1/G of the traffic and of the merge work of the replicated form.
The reference has no inference-time collective; its closest analogues are the training-time all_gather of
reps (GDR_model/encoder.py:134-145) and the offline per-GPU partitioning of Data_process/NQ_dataset/bert/bert.py:51-61.

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


class ShardedIndex:
    def __init__(self, D_shard, lo, group=None, local_topk=None, merge=None):
        self.D, self.lo, self.group = D_shard, int(lo), group
        if local_topk is None or merge is None:
            from . import ops
            ws = ops.Workspace(D_shard.device)
            local_topk = local_topk or (lambda Q, D, k, off: ops.sim_topk(Q, D, k, idx_offset=off, workspace=ws))
            merge = merge or ops.topk_merge
        self.local_topk, self.merge = local_topk, merge
        self.distributed = dist.is_initialized()      # a 1-rank group still runs the collectives (exercises RCCL)
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0

    def gather_queries(self, q_local):
        """[B_local,d] per rank -> [world*B_local,d], rank-major (every rank contributes the same count)."""
        if not self.distributed:
            return q_local
        out = torch.empty((self.world * q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype,
                          device=q_local.device)
        dist.all_gather_into_tensor(out, q_local.contiguous(), group=self.group)
        return out

    def search(self, q_all, k):
        """q_all [B,d] identical on every rank -> (values [B,k], global doc ids int32 [B,k]), identical on every rank."""
        v, i = self.local_topk(q_all, self.D, k, self.lo)
        if not self.distributed:
            return v, i
        B = q_all.shape[0]
        gv = torch.empty((self.world * B, k), dtype=v.dtype, device=v.device)     # rank-major concatenation
        gi = torch.empty((self.world * B, k), dtype=i.dtype, device=i.device)
        dist.all_gather_into_tensor(gv, v.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gi, i.contiguous(), group=self.group)
        return self.merge(gv.view(self.world, B, k), gi.view(self.world, B, k))

    def search_own(self, q_all, k):
        """q_all [B,d] identical on every rank (rank-major blocks of B/world queries) -> top-k of THIS rank's query block:
        (values [B/world,k], global doc ids int32 [B/world,k]).  Same lists as search()[rank block]."""
        v, i = self.local_topk(q_all, self.D, k, self.lo)
        if not self.distributed:
            return v, i
        B = q_all.shape[0]
        if B % self.world:
            raise ValueError("search_own: the gathered batch must hold the same number of queries per rank")
        bl = B // self.world
        rv = torch.empty_like(v)                       # [world*bl, k]: block g = rank g's list for my queries
        ri = torch.empty_like(i)
        dist.all_to_all_single(rv, v.contiguous(), group=self.group)
        dist.all_to_all_single(ri, i.contiguous(), group=self.group)
        return self.merge(rv.view(self.world, bl, k), ri.view(self.world, bl, k))
