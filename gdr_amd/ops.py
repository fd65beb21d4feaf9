"""Thin tensor-level wrappers over the C ABI (include/gdr_hip.h).  Inputs/outputs are torch CUDA tensors used
purely as device buffers; all arithmetic happens in libgdr_hip.so."""
import ctypes as C
import math

import torch

from . import _ffi
from ._ffi import check, lib, ptr, stream_ptr


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _ffi.GdrError("gdr_amd ops need CUDA (ROCm) tensors; there is no CPU path")


def _f32c(t):
    if t.dtype != torch.float32:
        raise _ffi.GdrError(f"expected float32 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def linear(a, w, epilogue=_ffi.EPI_NONE, bias=None, residual=None, out=None, splitk_ws=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K].T)  — gdr_linear_f32 (gdr_linear_f32_splitk when a scratch tensor is given)."""
    _need_cuda(a, w, bias, residual)
    K = a.shape[-1]
    a2 = _f32c(a).view(-1, K)
    w = _f32c(w)
    M, N = a2.shape[0], w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    res2 = None
    if residual is not None:
        res2 = _f32c(residual).view(-1, N)
    if splitk_ws is not None:
        check(lib().gdr_linear_f32_splitk(ptr(a2), K, ptr(w), w.shape[1], ptr(out), N, M, N, K, epilogue,
                                          ptr(_f32c(bias)) if bias is not None else None, ptr(res2), N, ptr(splitk_ws),
                                          splitk_ws.numel() * splitk_ws.element_size(), stream_ptr()),
              "gdr_linear_f32_splitk")
        return out.view(*a.shape[:-1], N)
    check(lib().gdr_linear_f32(ptr(a2), K, ptr(w), w.shape[1], ptr(out), N, M, N, K, epilogue,
                               ptr(_f32c(bias)) if bias is not None else None, ptr(res2), N, stream_ptr()),
          "gdr_linear_f32")
    return out.view(*a.shape[:-1], N)


def linear_bf16(a, w, epilogue=_ffi.EPI_NONE, bias=None, residual=None, out=None):
    """out[M,N] fp32 = epilogue(a[M,K] @ w[N,K].T) with bf16 operands (fp32 tensors are rounded on the device first) and
    fp32 accumulate — gdr_linear_bf16, the linear of the C5 precision mode."""
    _need_cuda(a, w, bias, residual)
    a = (a if a.dtype == torch.bfloat16 else to_bf16(a)).contiguous()
    w = (w if w.dtype == torch.bfloat16 else to_bf16(w)).contiguous()
    K = a.shape[-1]
    a2 = a.view(-1, K)
    M, N = a2.shape[0], w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    res2 = _f32c(residual).view(-1, N) if residual is not None else None
    check(lib().gdr_linear_bf16(ptr(a2), K, ptr(w), w.shape[1], ptr(out), N, M, N, K, epilogue,
                                ptr(_f32c(bias)) if bias is not None else None, ptr(res2), N, stream_ptr()),
          "gdr_linear_bf16")
    return out.view(*a.shape[:-1], N)


def split_bf16x3(x, padded=True):
    """fp32 [rows, K] -> bf16 [rows, ld] with ld = gdr_split_row_elems(K) >= 3K (padded=False: exactly 3K), a row = [hi | mid | lo | pad]
    with x = hi + mid + lo to 24 bits — gdr_split_f32_bf16x3."""
    _need_cuda(x)
    x = _f32c(x)
    K = x.shape[-1]
    ld = lib().gdr_split_row_elems(K, 6) if padded else 3 * K
    out = torch.zeros(x.shape[:-1] + (ld,), dtype=torch.bfloat16, device=x.device)
    check(lib().gdr_split_f32_bf16x3(ptr(x), ptr(out), x.numel() // K, K, ld, stream_ptr()), "gdr_split_f32_bf16x3")
    return out


def split_f16x2(x):
    """fp32 [rows, K] -> fp16 [rows, 2K], a row = [hi | lo'] with hi = fp16(x), lo' = fp16((x - hi) * 2^11): 22 bits — gdr_split_f32_f16x2."""
    _need_cuda(x)
    x = _f32c(x)
    K = x.shape[-1]
    ld = lib().gdr_split_row_elems(K, 2)
    out = torch.zeros(x.shape[:-1] + (ld,), dtype=torch.float16, device=x.device)
    check(lib().gdr_split_f32_f16x2(ptr(x), ptr(out), x.numel() // K, K, ld, stream_ptr()), "gdr_split_f32_f16x2")
    return out


def linear_split_bf16(a3, w3, K, epilogue=_ffi.EPI_NONE, bias=None, residual=None, out=None, terms=6):
    """out[M,N] fp32 = epilogue(a @ w.T) with a [M,K], w [N,K] given as three bf16 planes per row (split_bf16x3; rows may be padded
    beyond 3K) — gdr_linear_split_bf16: the six leading products of the 24-bit operands on the bf16 MFMA path, fp32 accumulate.
    Exploratory, beside ops.linear."""
    _need_cuda(a3, w3, bias, residual)
    want = torch.float16 if terms == 2 else torch.bfloat16
    planes = 2 if terms == 2 else 3
    if a3.dtype != want or w3.dtype != want or a3.shape[-1] < planes * K or w3.shape[-1] < planes * K:
        raise _ffi.GdrError("linear_split_bf16: operands must be plane rows (bf16 x 3 for terms 6 / 3, fp16 x 2 for terms 2) of the same K")
    a3, w3 = a3.contiguous(), w3.contiguous()
    a2 = a3.view(-1, a3.shape[-1])
    M, N = a2.shape[0], w3.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a3.device)
    res2 = _f32c(residual).view(-1, N) if residual is not None else None
    check(lib().gdr_linear_split_bf16(ptr(a2), a2.shape[1], ptr(w3), w3.shape[1], ptr(out), N, M, N, K, int(terms), epilogue,
                                      ptr(_f32c(bias)) if bias is not None else None, ptr(res2), N, stream_ptr()), "gdr_linear_split_bf16")
    return out.view(*a3.shape[:-1], N)


def l2_normalize(x, eps=1e-12):
    """x / max(||x||_2, eps) over the last dim — gdr_l2_normalize (torch.nn.functional.normalize, dense.py:24-25)."""
    _need_cuda(x)
    x = _f32c(x)
    out = torch.empty_like(x)
    d = x.shape[-1]
    check(lib().gdr_l2_normalize(ptr(x), ptr(out), x.numel() // d, d, float(eps), stream_ptr()), "gdr_l2_normalize")
    return out


def t5_layer_norm(x, weight, eps=1e-6):
    """T5LayerNorm (modeling_t5.py:164-171): weight * (x / sqrt(mean(x^2) + eps)) over the last dim, fp32 — gdr_t5_layer_norm."""
    _need_cuda(x, weight)
    x, weight = _f32c(x), _f32c(weight)
    d = x.shape[-1]
    if weight.numel() != d:
        raise _ffi.GdrError(f"t5_layer_norm: weight has {weight.numel()} elements, rows have {d}")
    out = torch.empty_like(x)
    check(lib().gdr_t5_layer_norm(ptr(x), ptr(weight), ptr(out), x.numel() // d, d, float(eps), stream_ptr()), "gdr_t5_layer_norm")
    return out


class Workspace:
    """Grow-only device scratch (256-byte aligned by the caching allocator), one buffer per HIP stream: calls that are in
    flight on different streams (GDRRetriever.validation_steps, two generate() calls) never share scratch."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, nbytes):
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        return buf


def to_bf16(x):
    """fp32 tensor -> bf16 tensor (round-to-nearest-even) — gdr_cast_f32_bf16."""
    _need_cuda(x)
    x = _f32c(x)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().gdr_cast_f32_bf16(ptr(x), ptr(out), x.numel(), stream_ptr()), "gdr_cast_f32_bf16")
    return out


def _sim_topk_raw(Q, D, k, idx_offset, workspace, flags):
    B, d = Q.shape
    N = D.shape[0]
    need = lib().gdr_sim_topk_workspace_bytes(B, N, d, k, flags)
    ws = (workspace or Workspace(Q.device)).get(need)
    vals = torch.empty((B, k), dtype=torch.float32, device=Q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=Q.device)
    status = torch.empty((B,), dtype=torch.int32, device=Q.device)
    fn = lib().gdr_sim_topk_bf16 if Q.dtype == torch.bfloat16 else lib().gdr_sim_topk
    check(fn(ptr(Q), B, ptr(D), N, d, k, idx_offset, ptr(vals), ptr(idx), ptr(status), flags, ptr(ws), ws.numel(),
             stream_ptr()), "gdr_sim_topk")
    return vals, idx, status


def sim_topk(Q, D, k, idx_offset=0, workspace=None, return_status=False, exact_on_overflow=True, flags=0):
    """Fused Q·Dᵀ + per-row top-k — gdr_sim_topk.  Returns (values fp32[B,k], indices int32[B,k]).
    exact_on_overflow=True (the default) reads the per-query status back (one host sync per call) and recomputes any
    query whose candidate list overflowed (degenerate corpora with tens of thousands of tied docs) exhaustively, so the
    result is exact for every input.  Latency-critical callers pass exact_on_overflow=False, return_status=True: no sync,
    and the device status tensor (1 = that row is the top-k of a subset) is theirs to act on.  flags: _ffi.SIM_* bits."""
    prefilter = isinstance(D, PrefilteredCorpus)
    if prefilter:
        P, D = D, D.D
        if (Q.shape[0] < PREFILTER_MIN_BATCH or D.shape[1] % 8 or D.shape[1] > 1024 or (flags & _ffi.SIM_EXHAUSTIVE)
                or not P.dnorm_max > 0.0):                               # an all-zero corpus has no band: the fp32 path serves it
            prefilter = False
    _need_cuda(Q, D)
    if D.dtype == torch.bfloat16:                                    # bf16 corpus: queries are cast on the device
        Q = (Q if Q.dtype == torch.bfloat16 else to_bf16(Q)).contiguous()
        D = D.contiguous()
    else:
        Q, D = _f32c(Q), _f32c(D)
    if D.shape[1] != Q.shape[1]:
        raise _ffi.GdrError(f"sim_topk: dim mismatch {Q.shape} vs {D.shape}")
    if k > D.shape[0]:
        raise RuntimeError("selected index k out of range")          # torch.topk's message
    if prefilter:
        vals, idx, status = _sim_topk_prefilter_raw(Q, P, k, idx_offset, workspace)
    else:
        vals, idx, status = _sim_topk_raw(Q, D, k, idx_offset, workspace, flags)
    if exact_on_overflow:
        bad = torch.nonzero(status).flatten()
        _ffi.check_device_fault("sim_topk")                              # nonzero() synchronised
        if bad.numel():
            for lo in range(0, bad.numel(), 64):                     # bounded workspace: 64 queries * N * 8 B
                rows = bad[lo:lo + 64]
                v2, i2, _ = _sim_topk_raw(Q[rows].contiguous(), D, k, idx_offset, None, _ffi.SIM_EXHAUSTIVE)
                vals[rows], idx[rows] = v2, i2
            status = torch.zeros_like(status)
    return (vals, idx, status) if return_status else (vals, idx)


class PrefilteredCorpus:
    """An fp32 corpus [N,d] with what gdr_sim_topk_prefilter needs beside it: its bf16 image (RNE, +50 % memory) and the largest
    row norm, both made once on the device.  ops.sim_topk(Q, PrefilteredCorpus(D), k) returns the top-k of the FP32 scores (the
    bf16 pass only decides which few hundred docs per query get one: include/gdr_hip.h).
    The image and the norm bound are SNAPSHOTS of D: after any in-place update of D (an index refresh) call refresh() before the
    next search — a stale image can drop docs from the band, a stale norm bound can make the band too narrow.  `.D` is the raw
    fp32 tensor for everything that is not a search (rerank, gathers).  An all-zero corpus (dnorm_max = 0) is served by the
    fp32 path."""

    def __init__(self, D):
        _need_cuda(D)
        self.D = _f32c(D)
        self.shape, self.dtype, self.device = self.D.shape, self.D.dtype, self.D.device
        self.D16 = torch.empty(self.D.shape, dtype=torch.bfloat16, device=self.D.device)
        self.refresh()

    def refresh(self):
        """Re-derive the bf16 image and the largest row norm from the current contents of D (one sync)."""
        check(lib().gdr_cast_f32_bf16(ptr(self.D), ptr(self.D16), self.D.numel(), stream_ptr()), "gdr_cast_f32_bf16")
        m = torch.empty(1, dtype=torch.float32, device=self.D.device)
        check(lib().gdr_row_norm2_max(ptr(self.D), self.D.shape[0], self.D.shape[1], ptr(m), stream_ptr()), "gdr_row_norm2_max")
        self.dnorm_max = float(m.item()) ** 0.5 * (1.0 + 1e-6)       # the squared norm is itself rounded
        return self


PREFILTER_MIN_BATCH = 1           # the pre-filter serves every batch size: at B <= 32 its corpus-wide pass is the HBM stream over
                                  # the bf16 image (half the bytes of the fp32 stream): 0.205 vs 0.241 ms at B = 1


def _sim_topk_prefilter_raw(Q, P, k, idx_offset, workspace):
    B, d = Q.shape
    N = P.D.shape[0]
    need = lib().gdr_sim_topk_prefilter_workspace_bytes(B, N, d, k)
    ws = (workspace or Workspace(Q.device)).get(need)
    vals = torch.empty((B, k), dtype=torch.float32, device=Q.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=Q.device)
    status = torch.empty((B,), dtype=torch.int32, device=Q.device)
    check(lib().gdr_sim_topk_prefilter(ptr(Q), B, ptr(P.D), ptr(P.D16), P.dnorm_max, N, d, k, idx_offset, ptr(vals), ptr(idx),
                                       ptr(status), ptr(ws), ws.numel(), stream_ptr()), "gdr_sim_topk_prefilter")
    return vals, idx, status


def topk_merge(vals, idx):
    """[G,B,k] per-shard lists -> [B,k] — gdr_topk_merge."""
    _need_cuda(vals, idx)
    vals, idx = _f32c(vals), idx.contiguous()
    G, B, k = vals.shape
    ov = torch.empty((B, k), dtype=torch.float32, device=vals.device)
    oi = torch.empty((B, k), dtype=torch.int32, device=vals.device)
    check(lib().gdr_topk_merge(ptr(vals), ptr(idx), G, B, k, ptr(ov), ptr(oi), stream_ptr()), "gdr_topk_merge")
    return ov, oi


def topk_pack(vals, idx, status=None):
    """(values fp32[B,k], ids int32[B,k][, status int32[B]]) -> int64[B,k+1] wire form — gdr_topk_pack."""
    _need_cuda(vals, idx, status)
    vals, idx = _f32c(vals), idx.contiguous()
    B, k = vals.shape
    pairs = torch.empty((B, k + 1), dtype=torch.int64, device=vals.device)
    check(lib().gdr_topk_pack(ptr(vals), ptr(idx), ptr(status), B, k, ptr(pairs), stream_ptr()), "gdr_topk_pack")
    return pairs


def topk_merge_packed(pairs, return_status=False):
    """int64[G,B,k+1] per-shard wire lists -> (values [B,k], ids int32 [B,k][, status int32 [B]]) — gdr_topk_merge_packed."""
    _need_cuda(pairs)
    pairs = pairs.contiguous()
    G, B, k1 = pairs.shape
    k = k1 - 1
    ov = torch.empty((B, k), dtype=torch.float32, device=pairs.device)
    oi = torch.empty((B, k), dtype=torch.int32, device=pairs.device)
    st = torch.empty((B,), dtype=torch.int32, device=pairs.device) if return_status else None
    check(lib().gdr_topk_merge_packed(ptr(pairs), G, B, k, ptr(ov), ptr(oi), ptr(st), stream_ptr()),
          "gdr_topk_merge_packed")
    return (ov, oi, st) if return_status else (ov, oi)


_RERANK_WS = {}


def rerank_topk(q, D, cand_offsets, cand_ids, beam_scores, alphas, k, func="tanh", max_cand=None, doc_range=None,
                positions=False, workspace=None, cand_stride=0):
    """In-cluster rerank — gdr_rerank_topk / gdr_rerank_topk_bf16 (chosen by D.dtype; a bf16 corpus is gathered as bf16,
    never up-cast).  Returns (values fp32[B,A,k], doc ids int32[B,A,k]).
    cand_stride=0: cand_offsets int32[B*R+1] is ONE CSR into cand_ids (the reference's concatenation order);
    cand_stride>0: per-query blocks — cand_offsets int32[B,R+1] relative, cand_ids int32[B,cand_stride] (DeviceClusterIndex).
    max_cand: bound of any query's candidate count; None reads it from the CSR (one host sync).
    doc_range=(lo, hi): D holds rows [lo, hi) of the corpus, candidates outside are skipped (sharded GDR mode, dist.py);
    positions=True returns candidate positions within the query's list instead of doc ids (what the shard merge needs)."""
    _need_cuda(q, D, cand_offsets, cand_ids, beam_scores)
    q, beam_scores = _f32c(q), _f32c(beam_scores)
    if D.dtype == torch.bfloat16:
        fn, D = lib().gdr_rerank_topk_bf16, D.contiguous()
    else:
        fn, D = lib().gdr_rerank_topk, _f32c(D)
    if D.dim() != 2 or D.shape[1] != q.shape[1]:
        raise _ffi.GdrError(f"rerank_topk: dim mismatch {tuple(q.shape)} vs {tuple(D.shape)}")
    B, R = beam_scores.shape
    al = torch.as_tensor(alphas, dtype=torch.float32, device=q.device)
    A = al.numel()
    if max_cand is None:
        if cand_stride:
            max_cand = int(cand_offsets.view(B, R + 1)[:, R].max().item())
        else:
            o = cand_offsets.view(-1)[::R]
            max_cand = int((o[1:] - o[:-1]).max().item())
    max_cand = max(int(max_cand), 1)
    if cand_stride and cand_stride < max_cand:
        raise _ffi.GdrError(f"rerank_topk: cand_stride {cand_stride} < max_cand {max_cand}")
    lo, hi = (0, D.shape[0]) if doc_range is None else (int(doc_range[0]), int(doc_range[1]))
    if hi - lo != D.shape[0]:
        raise _ffi.GdrError(f"rerank_topk: doc_range {(lo, hi)} does not match the {D.shape[0]} rows given")
    ws = (workspace or _RERANK_WS.setdefault(q.device, Workspace(q.device))).get(lib().gdr_rerank_workspace_bytes(B, max_cand))
    ov = torch.empty((B, A, k), dtype=torch.float32, device=q.device)
    oi = torch.empty((B, A, k), dtype=torch.int32, device=q.device)
    check(fn(ptr(q), ptr(D), q.shape[1], ptr(cand_offsets), ptr(cand_ids), ptr(beam_scores), B, R, ptr(al), A, k,
             0 if func == "tanh" else 1, ptr(ov), ptr(oi), max_cand, int(cand_stride), lo, hi,
             _ffi.RERANK_POSITIONS if positions else 0,
             ptr(ws), ws.numel(), stream_ptr()), "gdr_rerank_topk")
    return ov, oi


def rerank_wire_pack(q, beam_scores, cand_offsets, cand_ids):
    """(q fp32[B,d], beam fp32[B,R], offsets int32[B,R+1], ids int32[B,stride]) -> int32[B, d+2R+1+stride], the exchange row
    of the sharded in-cluster rerank — gdr_rerank_wire_pack."""
    _need_cuda(q, beam_scores, cand_offsets, cand_ids)
    q, beam_scores = _f32c(q), _f32c(beam_scores)
    cand_offsets, cand_ids = cand_offsets.to(torch.int32).contiguous(), cand_ids.to(torch.int32).contiguous()
    (B, d), R, stride = q.shape, beam_scores.shape[1], cand_ids.shape[1]
    wire = torch.empty((B, d + 2 * R + 1 + stride), dtype=torch.int32, device=q.device)
    check(lib().gdr_rerank_wire_pack(ptr(q), ptr(beam_scores), ptr(cand_offsets), ptr(cand_ids), B, d, R, stride, ptr(wire),
                                     stream_ptr()), "gdr_rerank_wire_pack")
    return wire


def rerank_wire_unpack(wire, d, R, stride):
    """int32[B, d+2R+1+stride] -> (q fp32[B,d], beam fp32[B,R], offsets int32[B,R+1], ids int32[B,stride]) — gdr_rerank_wire_unpack."""
    _need_cuda(wire)
    wire = wire.contiguous()
    B = wire.shape[0]
    if wire.dtype != torch.int32 or wire.shape[1] != d + 2 * R + 1 + stride:
        raise _ffi.GdrError(f"rerank_wire_unpack: wire {tuple(wire.shape)} {wire.dtype} does not match d={d} R={R} stride={stride}")
    q = torch.empty((B, d), dtype=torch.float32, device=wire.device)
    beam = torch.empty((B, R), dtype=torch.float32, device=wire.device)
    offs = torch.empty((B, R + 1), dtype=torch.int32, device=wire.device)
    ids = torch.empty((B, stride), dtype=torch.int32, device=wire.device)
    check(lib().gdr_rerank_wire_unpack(ptr(wire), B, d, R, stride, ptr(q), ptr(beam), ptr(offs), ptr(ids), stream_ptr()),
          "gdr_rerank_wire_unpack")
    return q, beam, offs, ids


def rerank_positions_to_ids(pos, cand_ids):
    """pos int32[B, ...] candidate positions (merged GDR_RERANK_POSITIONS lists; < 0 = padding) -> doc ids int32 of the same
    shape through cand_ids int32[B, stride] — gdr_rerank_positions_to_ids."""
    _need_cuda(pos, cand_ids)
    pos, cand_ids = pos.to(torch.int32).contiguous(), cand_ids.to(torch.int32).contiguous()
    B, stride = cand_ids.shape
    out = torch.empty_like(pos)
    check(lib().gdr_rerank_positions_to_ids(ptr(pos), ptr(cand_ids), B, pos.numel() // B, stride, ptr(out), stream_ptr()),
          "gdr_rerank_positions_to_ids")
    return out


RERANK_MAX_CAND = 8192          # csrc/rerank.hip RR_MAX_CAND: the LDS sort of one (alpha, query) list


def block_max_cand(cand_offsets, R, stride):
    """The `max_cand` to hand gdr_rerank_topk for candidate blocks of width `stride` = num beams x LARGEST cluster of the
    corpus.  While that worst case fits the kernel's cap it is used as is — nothing synchronises.  One outlier cluster
    (more than 81 docs at 100 beams) must not make every step fail when the clusters actually decoded are small: beyond
    the cap the bound comes from the data (one read-back of the per-query counts, this case only), and only a query that
    REALLY has more than 8192 candidates is refused (as the host CSR path refuses it)."""
    if stride <= RERANK_MAX_CAND:
        return max(int(stride), 1)
    real = int(cand_offsets.view(-1, R + 1)[:, R].max().item())
    if real > RERANK_MAX_CAND:
        raise _ffi.GdrError(f"rerank: a query decoded {real} candidate docs; the in-cluster rerank ranks at most "
                            f"{RERANK_MAX_CAND} per query")
    return max(real, 1)


class DeviceClusterIndex:
    """codec.ClusterIndex resident on the GPU (include/gdr_hip.h GdrClusterIndex): the clusters' token bodies in an
    exact-match hash table + the member CSR, so that gdr_cluster_candidates turns generate()'s output rows into the rerank's
    candidate CSR without a host round trip (main_models.py:1398,1441-1443)."""

    def __init__(self, index, device, V, position=1, kary=30):
        import numpy as np
        bodies = index.token_bodies(V, position=position, kary=kary)       # list of int lists, None = unreachable name
        n = len(index.names)
        key_len = max([len(b) for b in bodies if b is not None] + [1])
        keys = np.full((max(n, 1), key_len), -1, np.int32)
        lens = np.full(max(n, 1), -1, np.int32)
        T = 4
        while T < 2 * max(n, 1):
            T <<= 1
        slots = np.full(T, -1, np.int32)
        hfn = lib().gdr_cluster_key_hash
        where = {}
        for c, b in enumerate(bodies):
            if b is None:
                continue
            where[tuple(b)] = c                                            # a repeated name: the last one wins, like the dict
        for b, c in where.items():
            keys[c, :len(b)] = b
            lens[c] = len(b)
            arr = (C.c_int32 * max(len(b), 1))(*b)
            slot = int(hfn(arr, len(b))) & (T - 1)
            while slots[slot] >= 0:
                slot = (slot + 1) & (T - 1)
            slots[slot] = c
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
        self.slots, self.keys, self.lens = dev(slots), dev(keys), dev(lens)
        self.offsets, self.members = dev(index.offsets.astype(np.int32)), dev(index.members.astype(np.int32) if index.members.size
                                                                             else np.zeros(1, np.int32))
        self.max_cluster = int(np.diff(index.offsets).max()) if n else 0
        self.struct = _ffi.GdrClusterIndex(n, key_len, T, self.slots.data_ptr(), self.keys.data_ptr(), self.lens.data_ptr(),
                                           self.offsets.data_ptr(), self.members.data_ptr())

    def candidates(self, out_ids, B, R):
        """out_ids int64[B*R, max_length] (device, untrimmed) -> (cluster_of int32[B*R], cand_offsets int32[B,R+1],
        cand_ids int32[B,stride], stride = R * largest cluster) — the per-query block layout of rerank_topk
        (cand_stride=stride, max_cand=stride); all on the device, nothing synchronises."""
        _need_cuda(out_ids)
        out_ids = out_ids.contiguous()
        stride = max(R * self.max_cluster, 1)
        cl = torch.empty((B * R,), dtype=torch.int32, device=out_ids.device)
        offs = torch.empty((B, R + 1), dtype=torch.int32, device=out_ids.device)
        ids = torch.empty((B, stride), dtype=torch.int32, device=out_ids.device)
        check(lib().gdr_cluster_candidates(C.byref(self.struct), ptr(out_ids), B, R, out_ids.shape[1], ptr(cl), ptr(offs),
                                           ptr(ids), stride, stream_ptr()), "gdr_cluster_candidates")
        return cl, offs, ids, stride


def relative_bucket_table(bidirectional, num_buckets, max_distance, qlen, klen):
    """Host table int32[qlen,klen] from the same routine the attention kernels use (no GPU needed)."""
    buf = (C.c_int32 * (qlen * klen))()
    check(lib().gdr_t5_relative_bucket_table(int(bidirectional), num_buckets, max_distance, qlen, klen, buf),
          "gdr_t5_relative_bucket_table")
    return torch.tensor(list(buf), dtype=torch.int32).view(qlen, klen)


class T5EncoderHandle:
    """Device-resident encoder weights + the pointer table gdr_t5_encoder_forward reads.
    Built once from a reference-style state_dict; q/k/v are row-concatenated.
    dtype=torch.bfloat16 selects the C5 precision mode: the linear weights are rounded to bf16 on the device
    (gdr_cast_f32_bf16) and forward() calls gdr_t5_encoder_forward_bf16; everything else stays fp32."""

    def __init__(self, cfg, sd, device, prefix="encoder.", dtype=torch.float32, split=False):
        """split=True (r06, exploratory; dtype float32): every linear weight is stored as three bf16 planes (split_bf16x3) and forward()
        runs gdr_t5_encoder_forward_ragged_split — fp32 operands carried through bf16 MFMAs with fp32-level error, not fp32 bits."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("T5EncoderHandle: dtype must be float32 or bfloat16")
        if split and dtype != torch.float32:
            raise ValueError("T5EncoderHandle: split=True is a form of the float32 mode")
        if split not in (False, True, 0, 2, 3, 6):
            raise ValueError("T5EncoderHandle: split must be False, True (= 6 terms), 3 / 6 (bf16 planes) or 2 (fp16 x 2)")
        self.cfg, self.device, self.dtype, self.split = cfg, device, dtype, (6 if split is True else int(split))
        keep = []

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        def lin(t):                                      # a linear's weight: bf16 copy in the C5 mode, three bf16 planes in the split form
            t = dev(t)
            if dtype == torch.bfloat16:
                keep.pop()
                t = to_bf16(t)
                keep.append(t)
            elif split:
                keep.pop()
                t = split_f16x2(t) if int(split) == 2 else split_bf16x3(t)
                keep.append(t)
            return t

        self.embed = dev(sd["shared.weight"] if "shared.weight" in sd else sd[prefix + "embed_tokens.weight"])
        self.rel_bias = dev(sd[prefix + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"])
        self.final_ln = dev(sd[prefix + "final_layer_norm.weight"])
        nl = cfg.num_layers
        self._layers = (_ffi.GdrT5EncLayer * nl)()
        for i in range(nl):
            p = f"{prefix}block.{i}.layer."
            wqkv = lin(torch.cat([sd[p + "0.SelfAttention.q.weight"], sd[p + "0.SelfAttention.k.weight"],
                                  sd[p + "0.SelfAttention.v.weight"]], dim=0))
            L = self._layers[i]
            L.ln_attn = dev(sd[p + "0.layer_norm.weight"]).data_ptr()
            L.wqkv = wqkv.data_ptr()
            L.wo = lin(sd[p + "0.SelfAttention.o.weight"]).data_ptr()
            L.ln_ff = dev(sd[p + "1.layer_norm.weight"]).data_ptr()
            L.wi = lin(sd[p + "1.DenseReluDense.wi.weight"]).data_ptr()
            L.wo_ff = lin(sd[p + "1.DenseReluDense.wo.weight"]).data_ptr()
        self._keep = keep
        self.dims = _ffi.GdrT5Dims(cfg.vocab_size, cfg.d_model, cfg.d_kv, cfg.d_ff, cfg.num_heads, nl,
                                   cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance,
                                   cfg.layer_norm_epsilon)
        self.struct = _ffi.GdrT5EncoderWeights(self.dims, self.embed.data_ptr(), self.rel_bias.data_ptr(),
                                               self.final_ln.data_ptr(), self._layers)
        self.ws = Workspace(device)

    def forward(self, input_ids, attention_mask=None, want_pooled=True, want_hidden=True, ragged=False,
                live_rows_hint=-1):
        """Returns (last_hidden_state fp32[B,L,d] | None, pooled fp32[B,d] | None).
        ragged=False: gdr_t5_encoder_forward — every row, PAD positions included, exactly as the reference computes them.
        ragged=True : gdr_t5_encoder_forward_ragged — PAD rows are not computed (kept rows bit-identical, PAD rows of the
        returned hidden states are zero); with want_hidden=False only h[:,0] is carried through the last block.
        live_rows_hint: number of kept token rows if the caller knows it — a tuning input (the launcher chooses between
        bit-identical kernel forms by the tile count it implies; the profiler prices flops with it), never a result input."""
        _need_cuda(input_ids, attention_mask)
        ids = input_ids.to(torch.int64).contiguous()
        B, L = ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(ids)
        mask = attention_mask.to(torch.int64).contiguous()
        bf = self.dtype == torch.bfloat16
        if not (want_hidden or want_pooled):
            raise ValueError("T5EncoderHandle.forward: nothing requested")
        d = self.cfg.d_model
        pooled = torch.empty((B, d), dtype=torch.float32, device=ids.device) if want_pooled else None
        if self.split:
            if not ragged:
                raise _ffi.GdrError("T5EncoderHandle(split=True): the split form exists for the ragged forward only")
            need = lib().gdr_t5_encoder_split_workspace_bytes(C.byref(self.dims), B, L)
            ws = self.ws.get(need)
            out = torch.empty((B, L, d), dtype=torch.float32, device=ids.device) if want_hidden else None
            check(lib().gdr_t5_encoder_forward_ragged_split(C.byref(self.struct), ptr(ids), ptr(mask), B, L, ptr(out), ptr(pooled),
                                                            int(live_rows_hint), self.split, ptr(ws), ws.numel(), stream_ptr()),
                  "gdr_t5_encoder_forward_ragged_split")
            return out, pooled
        if ragged:
            need = lib().gdr_t5_encoder_ragged_workspace_bytes(C.byref(self.dims), B, L)
            ws = self.ws.get(need)
            out = torch.empty((B, L, d), dtype=torch.float32, device=ids.device) if want_hidden else None
            rfn = lib().gdr_t5_encoder_forward_ragged_bf16 if bf else lib().gdr_t5_encoder_forward_ragged
            check(rfn(C.byref(self.struct), ptr(ids), ptr(mask), B, L, ptr(out), ptr(pooled), int(live_rows_hint), ptr(ws),
                      ws.numel(), stream_ptr()), "gdr_t5_encoder_forward_ragged")
            return out, pooled
        need = (lib().gdr_t5_encoder_bf16_workspace_bytes if bf else lib().gdr_t5_encoder_workspace_bytes)(
            C.byref(self.dims), B, L)
        ws = self.ws.get(need)
        out = torch.empty((B, L, d), dtype=torch.float32, device=ids.device)
        fn = lib().gdr_t5_encoder_forward_bf16 if bf else lib().gdr_t5_encoder_forward
        check(fn(C.byref(self.struct), ptr(ids), ptr(mask), B, L, ptr(out), ptr(pooled), ptr(ws), ws.numel(),
                 stream_ptr()), "gdr_t5_encoder_forward")
        return (out if want_hidden else None), pooled


class BertEncoderHandle:
    """Device-resident doc-tower weights (DPRContextEncoder / BertModel keys) + pointer table.
    dtype=torch.bfloat16 selects the bf16 precision mode (config C5's corpus is bf16): the linear weights are rounded to bf16 on the
    device, the attention's 1/sqrt(dh) is folded into the q rows of wqkv / bqkv (exact: a power of two at dh = 64), and forward()
    runs gdr_bert_encoder_forward_ragged_bf16 — the packed form is the only bf16 form."""

    def __init__(self, bcfg, sd, device, prefix="ctx_encoder.bert_model.", dtype=torch.float32, split=False):
        """split=True (r06, exploratory; dtype float32): the linear weights are stored as fp16 x 2 plane rows (split_f16x2) and forward()
        runs gdr_bert_encoder_forward_ragged_split — fp32-level embeddings through the fp16 MFMA path."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("BertEncoderHandle: dtype must be float32 or bfloat16")
        if split and dtype != torch.float32:
            raise ValueError("BertEncoderHandle: split=True is a form of the float32 mode")
        self.bcfg, self.device, self.dtype, self.split = bcfg, device, dtype, bool(split)
        bf = dtype == torch.bfloat16
        keep = []

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        def lin(t):                                      # a linear's weight: bf16 copy in the bf16 mode
            t = dev(t)
            if bf:
                keep.pop()
                t = to_bf16(t)
                keep.append(t)
            elif split:
                keep.pop()
                t = split_f16x2(t)
                keep.append(t)
            return t

        d, H = bcfg["hidden_size"], bcfg["num_heads"]
        qs = 1.0
        if bf:
            dh = d // H
            qs = float(dh) ** -0.5
            if d % H or 2.0 ** round(math.log2(qs)) != qs:
                raise _ffi.GdrError(f"BertEncoderHandle(bf16): head width {dh} — the scale fold needs 1/sqrt(dh) to be a power of two")
        e = prefix + "embeddings."
        self.word, self.pos = dev(sd[e + "word_embeddings.weight"]), dev(sd[e + "position_embeddings.weight"])
        self.type = dev(sd[e + "token_type_embeddings.weight"])
        self.eln_w, self.eln_b = dev(sd[e + "LayerNorm.weight"]), dev(sd[e + "LayerNorm.bias"])
        nl = bcfg["num_layers"]
        self._layers = (_ffi.GdrBertLayer * nl)()
        for i in range(nl):
            p = f"{prefix}encoder.layer.{i}."
            L = self._layers[i]
            wq, bq = sd[p + "attention.self.query.weight"] * qs, sd[p + "attention.self.query.bias"] * qs
            L.wqkv = lin(torch.cat([wq] + [sd[p + f"attention.self.{n}.weight"] for n in ("key", "value")], 0)).data_ptr()
            L.bqkv = dev(torch.cat([bq] + [sd[p + f"attention.self.{n}.bias"] for n in ("key", "value")], 0)).data_ptr()
            L.wo, L.bo = lin(sd[p + "attention.output.dense.weight"]).data_ptr(), dev(sd[p + "attention.output.dense.bias"]).data_ptr()
            L.ln1_w = dev(sd[p + "attention.output.LayerNorm.weight"]).data_ptr()
            L.ln1_b = dev(sd[p + "attention.output.LayerNorm.bias"]).data_ptr()
            L.wi, L.bi = lin(sd[p + "intermediate.dense.weight"]).data_ptr(), dev(sd[p + "intermediate.dense.bias"]).data_ptr()
            L.wo2, L.bo2 = lin(sd[p + "output.dense.weight"]).data_ptr(), dev(sd[p + "output.dense.bias"]).data_ptr()
            L.ln2_w, L.ln2_b = dev(sd[p + "output.LayerNorm.weight"]).data_ptr(), dev(sd[p + "output.LayerNorm.bias"]).data_ptr()
        self._keep = keep
        self.struct = _ffi.GdrBertWeights(self.word.shape[0], bcfg["hidden_size"], bcfg["num_heads"], bcfg["d_ff"], nl,
                                          self.pos.shape[0], self.type.shape[0], bcfg["eps"], self.word.data_ptr(),
                                          self.pos.data_ptr(), self.type.data_ptr(), self.eln_w.data_ptr(),
                                          self.eln_b.data_ptr(), self._layers)
        self.ws = Workspace(device)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, want_hidden=True, ragged=None, live_rows_hint=-1):
        """Returns (sequence_output fp32[B,L,d] | None, pooled fp32[B,d]).
        ragged=False: gdr_bert_encoder_forward — every position of every passage, as the reference computes them.
        ragged=True : gdr_bert_encoder_forward_ragged — PAD rows are not computed (kept rows bit-identical, PAD rows of the returned
        hidden states zero); with want_hidden=False only the CLS rows go through the last block.  Default: ragged for the bf16
        mode (its only form), padded otherwise."""
        _need_cuda(input_ids, attention_mask, token_type_ids)
        bf = self.dtype == torch.bfloat16
        ragged = (bf or self.split) if ragged is None else ragged
        if (bf or self.split) and not ragged:
            raise _ffi.GdrError("BertEncoderHandle: the bf16 precision mode and the split form exist in the packed (ragged) form only")
        ids = input_ids.to(torch.int64).contiguous()
        B, L = ids.shape
        mask = (torch.ones_like(ids) if attention_mask is None else attention_mask.to(torch.int64)).contiguous()
        tt = None if token_type_ids is None else token_type_ids.to(torch.int64).contiguous()
        d = self.bcfg["hidden_size"]
        hid = torch.empty((B, L, d), dtype=torch.float32, device=ids.device) if want_hidden else None
        pooled = torch.empty((B, d), dtype=torch.float32, device=ids.device)
        if ragged:
            need = lib().gdr_bert_encoder_ragged_workspace_bytes(C.byref(self.struct), B, L)
            ws = self.ws.get(need)
            fn = (lib().gdr_bert_encoder_forward_ragged_bf16 if bf else lib().gdr_bert_encoder_forward_ragged_split if self.split
                  else lib().gdr_bert_encoder_forward_ragged)
            check(fn(C.byref(self.struct), ptr(ids), ptr(mask), ptr(tt), B, L, ptr(hid), ptr(pooled), int(live_rows_hint), ptr(ws),
                     ws.numel(), stream_ptr()), "gdr_bert_encoder_forward_ragged")
            return hid, pooled
        need = lib().gdr_bert_encoder_workspace_bytes(C.byref(self.struct), B, L)
        ws = self.ws.get(need)
        check(lib().gdr_bert_encoder_forward(C.byref(self.struct), ptr(ids), ptr(mask), ptr(tt), B, L, ptr(hid), ptr(pooled),
                                             ptr(ws), ws.numel(), stream_ptr()), "gdr_bert_encoder_forward")
        return hid, pooled


class T5DecoderHandle:
    """Device-resident decoder + adaptor + head weights and the pointer table gdr_t5_generate reads.
    Load-time re-layouts (pure data movement / weight-only algebra, done once):
      * q,k,v of self-attention row-concatenated; k,v of cross-attention row-concatenated;
      * adaptor_linear.weight [d*Vd, d] (= [i, c, k], modeling_t5.py:1634-1636) sliced per decode position into
        head_w[p][c'][i][k] for the V+1 columns that survive the positional mask (modeling_t5.py:1553-1557);
      * the adaptor's cross-attention over its single learned key folded into cross_const (softmax of one key = 1)."""

    def __init__(self, cfg, sd, device, dtype=torch.float32):
        """dtype=torch.bfloat16: the C5 precision mode — every linear weight (decoder, adaptor, head slices) is rounded to
        bf16 on the device and generate() calls gdr_t5_generate_bf16; everything else stays fp32."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("T5DecoderHandle: dtype must be float32 or bfloat16")
        self.cfg, self.device, self.dtype = cfg, device, dtype
        keep = []

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        def lin(t):                                      # a linear's weight: bf16 copy in the C5 mode
            t = dev(t)
            if dtype == torch.bfloat16:
                keep.pop()
                t = to_bf16(t)
                keep.append(t)
            return t

        d, V, Vd, ml = cfg.d_model, cfg.output_vocab_size, cfg.decode_vocab_size, cfg.max_output_length
        nl, na = cfg.num_decoder_layers, cfg.adaptor_layer_num
        self.dec_embed = dev(sd["decode_embeddings.weight"])
        self.self_rel = dev(sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"])
        self.cross_rel = dev(sd["decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight"])
        self.final_ln = dev(sd["decoder.final_layer_norm.weight"])
        self._layers = (_ffi.GdrT5DecLayer * nl)()
        for i in range(nl):
            p = f"decoder.block.{i}.layer."
            L = self._layers[i]
            L.ln_self = dev(sd[p + "0.layer_norm.weight"]).data_ptr()
            L.wqkv = lin(torch.cat([sd[p + "0.SelfAttention.q.weight"], sd[p + "0.SelfAttention.k.weight"],
                                    sd[p + "0.SelfAttention.v.weight"]], dim=0)).data_ptr()
            L.wo = lin(sd[p + "0.SelfAttention.o.weight"]).data_ptr()
            L.ln_cross = dev(sd[p + "1.layer_norm.weight"]).data_ptr()
            L.wq_c = lin(sd[p + "1.EncDecAttention.q.weight"]).data_ptr()
            L.wkv_c = lin(torch.cat([sd[p + "1.EncDecAttention.k.weight"], sd[p + "1.EncDecAttention.v.weight"]],
                                    dim=0)).data_ptr()
            L.wo_c = lin(sd[p + "1.EncDecAttention.o.weight"]).data_ptr()
            L.ln_ff = dev(sd[p + "2.layer_norm.weight"]).data_ptr()
            L.wi = lin(sd[p + "2.DenseReluDense.wi.weight"]).data_ptr()
            L.wo_ff = lin(sd[p + "2.DenseReluDense.wo.weight"]).data_ptr()
        mem = dev(sd["adaptor_embeddings"]).view(1, d)
        self._alayers = (_ffi.GdrAdaptorLayer * na)()
        aff = None
        for i in range(na):
            p = f"adaptor.layers.{i}."
            A = self._alayers[i]
            A.in_w = lin(sd[p + "self_attn.in_proj_weight"]).data_ptr()
            A.in_b = dev(sd[p + "self_attn.in_proj_bias"]).data_ptr()
            A.out_w = lin(sd[p + "self_attn.out_proj.weight"]).data_ptr()
            A.out_b = dev(sd[p + "self_attn.out_proj.bias"]).data_ptr()
            cw, cb = dev(sd[p + "multihead_attn.in_proj_weight"]), dev(sd[p + "multihead_attn.in_proj_bias"])
            flin = linear_bf16 if dtype == torch.bfloat16 else linear        # the two folded linears round like all others
            vmem = flin(mem, cw[2 * d:].contiguous(), epilogue=_ffi.EPI_BIAS, bias=cb[2 * d:].contiguous())
            cc = flin(vmem, dev(sd[p + "multihead_attn.out_proj.weight"]), epilogue=_ffi.EPI_BIAS,
                      bias=dev(sd[p + "multihead_attn.out_proj.bias"]))
            keep.append(cc)
            A.cross_const = cc.data_ptr()
            for n in ("1", "2", "3"):
                setattr(A, f"ln{n}_w", dev(sd[p + f"norm{n}.weight"]).data_ptr())
                setattr(A, f"ln{n}_b", dev(sd[p + f"norm{n}.bias"]).data_ptr())
            l1 = lin(sd[p + "linear1.weight"])
            aff = l1.shape[0]
            A.lin1_w, A.lin1_b = l1.data_ptr(), dev(sd[p + "linear1.bias"]).data_ptr()
            A.lin2_w, A.lin2_b = lin(sd[p + "linear2.weight"]).data_ptr(), dev(sd[p + "linear2.bias"]).data_ptr()
        # head slices
        P = ml - 1
        cols = torch.tensor([[p * V + 2 + c for c in range(V)] + [1] for p in range(P)], dtype=torch.long, device=device)
        Wfull = sd["adaptor_linear.weight"].detach().to(device=device, dtype=torch.float32)
        W = Wfull.view(d, Vd, d)
        self.head_w = torch.empty((P, V + 1, d, d), dtype=dtype, device=device)
        for p in range(P):                                   # per position: bounded temporaries
            sl = W[:, cols[p], :].permute(1, 0, 2).contiguous()
            self.head_w[p] = to_bf16(sl) if dtype == torch.bfloat16 else sl
        del W, Wfull                                         # only the slices stay resident
        self.head_e = dev(sd["lm_head.weight"])[cols].contiguous()            # [P, V+1, d]
        self._keep = keep
        self.dims = _ffi.GdrT5Dims(Vd, d, cfg.d_kv, cfg.d_ff, cfg.num_heads, nl, cfg.relative_attention_num_buckets,
                                   cfg.relative_attention_max_distance, cfg.layer_norm_epsilon)
        self.struct = _ffi.GdrT5DecoderWeights(self.dims, V, ml, na, cfg.adaptor_nhead, aff, cfg.adaptor_ln_eps,
                                               self.dec_embed.data_ptr(), self.self_rel.data_ptr(),
                                               self.cross_rel.data_ptr(), self.final_ln.data_ptr(), self._layers,
                                               self._alayers, self.head_w.data_ptr(), self.head_e.data_ptr())
        self.ws = Workspace(device)

    def generate(self, enc_hidden, enc_mask, num_beams, max_length, length_penalty, num_return_sequences, trace=False,
                 trie=None, prefix_table=None, graph=False):
        """Returns (out_ids int64[B*nret,max_length], out_len int32[B*nret], out_scores float64[B*nret][, trace]).
        graph=True: the ~1 400 kernel launches of one call are captured once per (B, L, beams, max_length, nret) into a HIP
        graph (gdr_t5_generate contains no host synchronisation and forks / joins its side stream with events, i.e. it is
        capturable as is) and replayed on later calls — at small batches (one query x 100 beams, infer.sh's setting) the
        call is otherwise bound by the host's launch rate, not by the GPU."""
        _need_cuda(enc_hidden, enc_mask)
        enc_hidden = _f32c(enc_hidden)
        mask = enc_mask.to(torch.int64).contiguous()
        B, L, _ = enc_hidden.shape
        R, nret = int(num_beams), int(num_return_sequences)
        dev_ = enc_hidden.device
        fn = lib().gdr_t5_generate_bf16 if self.dtype == torch.bfloat16 else lib().gdr_t5_generate
        need = lib().gdr_t5_generate_workspace_bytes(C.byref(self.struct), B, L, R, max_length)

        def call(enc_t, mask_t, ids, lens, scores, ts, tt, ws):
            check(fn(C.byref(self.struct), ptr(enc_t), ptr(mask_t), B, L, R, max_length,
                     float(length_penalty), nret, trie.struct_ref() if trie is not None else None,
                     prefix_table.struct_ref() if prefix_table is not None else None,
                     ptr(ids), ptr(lens), ptr(scores), ptr(ts), ptr(tt),
                     ptr(ws), ws.numel(), stream_ptr()), "gdr_t5_generate")

        def outputs():
            ids = torch.empty((B * nret, max_length), dtype=torch.int64, device=dev_)
            lens = torch.empty((B * nret,), dtype=torch.int32, device=dev_)
            scores = torch.empty((B * nret,), dtype=torch.float64, device=dev_)
            return ids, lens, scores

        if graph and not trace:
            # one graph (and one set of static buffers) per call shape AND per stream: calls in flight on different
            # streams (GDRRetriever.validation_steps) replay different instances
            key = (B, L, R, max_length, nret, float(length_penalty), id(trie), id(prefix_table),
                   torch.cuda.current_stream(dev_).cuda_stream)
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            entry = self._graphs.get(key)
            if entry is None:
                s_enc, s_mask = torch.empty_like(enc_hidden), torch.empty_like(mask)
                s_ids, s_lens, s_scores = outputs()
                s_ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev_)
                s_enc.copy_(enc_hidden)
                s_mask.copy_(mask)
                warm = torch.cuda.Stream(device=dev_)                 # capture needs one eager run first (kernel
                warm.wait_stream(torch.cuda.current_stream())         # attributes, the side-stream lease, lazy module loads)
                with torch.cuda.stream(warm):
                    call(s_enc, s_mask, s_ids, s_lens, s_scores, None, None, s_ws)
                torch.cuda.current_stream().wait_stream(warm)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    call(s_enc, s_mask, s_ids, s_lens, s_scores, None, None, s_ws)
                entry = self._graphs[key] = (g, s_enc, s_mask, s_ids, s_lens, s_scores, s_ws)
            g, s_enc, s_mask, s_ids, s_lens, s_scores, _ = entry
            s_enc.copy_(enc_hidden)
            s_mask.copy_(mask)
            g.replay()
            return s_ids.clone(), s_lens.clone(), s_scores.clone()

        ws = self.ws.get(need)
        ids, lens, scores = outputs()
        ts = tt = None
        if trace:
            ts = torch.empty((max_length - 1, B, 2 * R), dtype=torch.float32, device=dev_)
            tt = torch.empty((max_length - 1, B, 2 * R), dtype=torch.int32, device=dev_)
        call(enc_hidden, mask, ids, lens, scores, ts, tt, ws)
        return (ids, lens, scores, ts, tt) if trace else (ids, lens, scores)


class DeviceTrie:
    """codec.Trie arrays resident on the GPU + the GdrTrie struct."""

    def __init__(self, trie, device):
        self.child = torch.from_numpy(trie.child).to(device).contiguous()
        self.eos_ok = torch.from_numpy(trie.eos_ok).to(device).contiguous()
        self.struct = _ffi.GdrTrie(self.child.data_ptr(), self.eos_ok.data_ptr(), self.child.shape[0], int(trie.V))

    def struct_ref(self):
        return C.byref(self.struct)


class PrefixTable:
    """Device prefix table (include/gdr_hip.h GdrPrefixTable) over a docid trie, built once from the decoder handle's
    weights by gdr_t5_prefix_table_build: per trie node the adaptor's per-layer (q,k,v) and the head matrix
    W = adaptor_linear(adaptor(prefix)) + lm_head of the node's position (modeling_t5.py:1618-1639, query-independent).
    `trie` is a codec.Trie in any order; `.device_trie` is its breadth-first DeviceTrie — pass THAT one as the constraint
    trie when both are used (gdr_t5_generate checks that they share their arrays)."""

    def __init__(self, dec, trie, device, max_levels=None, max_bytes=None):
        """max_levels: cap on the trie depth stored.  max_bytes: HBM budget of the table (default: half of what is free on
        the device now) — the table costs n_table * (adaptor_layers*3*d + (V+1)*d) * 4 bytes (~130 KB per node at t5-base),
        so the deepest levels are dropped until it fits (rows whose prefix is deeper simply take the computed path);
        a table that does not even hold the root level raises."""
        import numpy as np
        bfs, level_off, parent, tok = trie.breadth_first()
        cfg = dec.cfg
        n_levels = min(len(level_off) - 1, cfg.max_output_length - 1, max_levels or (1 << 30))
        per_node = (cfg.adaptor_layer_num * 3 * cfg.d_model + (cfg.output_vocab_size + 1) * cfg.d_model) * 4
        if max_bytes is None:
            max_bytes = torch.cuda.mem_get_info(device)[0] // 2
        full = n_levels
        while n_levels > 1 and int(level_off[n_levels]) * per_node > max_bytes:
            n_levels -= 1
        if int(level_off[n_levels]) * per_node > max_bytes:
            raise _ffi.GdrError(f"PrefixTable: even {int(level_off[n_levels])} nodes need more than the {max_bytes >> 20} MiB budget")
        if n_levels < full:
            print(f"[gdr_amd] prefix table: {int(level_off[full])} trie nodes would need "
                  f"{int(level_off[full]) * per_node / 2**30:.1f} GiB; keeping the first {n_levels} of {full} levels "
                  f"({int(level_off[n_levels])} nodes, {int(level_off[n_levels]) * per_node / 2**30:.2f} GiB) — deeper prefixes are computed")
        n_table = int(level_off[n_levels])
        self.device_trie = DeviceTrie(bfs, device)
        anc_blocks, anc = [], np.zeros((1, 1), np.int32)                 # level 0: the root's ancestor list is itself
        for s in range(n_levels):
            lo, hi = int(level_off[s]), int(level_off[s + 1])
            if s > 0:
                anc = np.concatenate([anc[parent[lo:hi] - int(level_off[s - 1])], np.arange(lo, hi, dtype=np.int32)[:, None]], 1)
            anc_blocks.append(anc.reshape(-1))
        node_anc = torch.from_numpy(np.concatenate(anc_blocks).astype(np.int32)).to(device)
        node_tok = torch.from_numpy(tok[:n_table].copy()).to(device)
        d, V1, na = cfg.d_model, cfg.output_vocab_size + 1, cfg.adaptor_layer_num
        self.kv = torch.empty((na, n_table, 3 * d), dtype=torch.float32, device=device)
        self.W = torch.empty((n_table, V1, d), dtype=torch.float32, device=device)
        lo_host = (C.c_int32 * (n_levels + 1))(*[int(x) for x in level_off[:n_levels + 1]])
        max_n = int(max(level_off[s + 1] - level_off[s] for s in range(n_levels)))
        need = lib().gdr_t5_prefix_table_workspace_bytes(C.byref(dec.struct), max_n)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        build = lib().gdr_t5_prefix_table_build_bf16 if dec.dtype == torch.bfloat16 else lib().gdr_t5_prefix_table_build
        check(build(C.byref(dec.struct), n_levels, lo_host, ptr(node_tok), ptr(node_anc), ptr(self.kv), ptr(self.W), ptr(ws),
                    ws.numel(), stream_ptr()), "gdr_t5_prefix_table_build")
        torch.cuda.current_stream().synchronize()                        # the scratch tensors above may go now
        self.n_levels, self.n_table, self.level_off = n_levels, n_table, level_off
        # levels 0 .. c-1 hold ALL V^s prefixes of their length (and are inside the table): no beam row can miss at those steps
        c = 1
        while c < n_levels and int(level_off[c + 1]) - int(level_off[c]) == int(bfs.V) ** c:
            c += 1
        self.complete_levels = c
        self.struct = _ffi.GdrPrefixTable(self.device_trie.child.data_ptr(), self.device_trie.child.shape[0], int(bfs.V),
                                          n_table, self.kv.data_ptr(), self.W.data_ptr(), c)

    def struct_ref(self):
        return C.byref(self.struct)

    def nbytes(self):
        return self.kv.numel() * 4 + self.W.numel() * 4


def beam_search_table(table, out_vocab, num_beams, max_length, length_penalty, num_return_sequences=None, trie=None):
    """Device beam search driven by a logit table [B, max_length, Vd, Vd] — gdr_beam_search_table."""
    _need_cuda(table)
    table = _f32c(table)
    B = table.shape[0]
    nret = num_return_sequences or num_beams
    need = lib().gdr_beam_search_table_workspace_bytes(B, num_beams, max_length, out_vocab)
    ws = torch.empty(need, dtype=torch.uint8, device=table.device)
    ids = torch.empty((B * nret, max_length), dtype=torch.int64, device=table.device)
    lens = torch.empty((B * nret,), dtype=torch.int32, device=table.device)
    scores = torch.empty((B * nret,), dtype=torch.float64, device=table.device)
    check(lib().gdr_beam_search_table(ptr(table), B, out_vocab, num_beams, max_length, float(length_penalty), nret,
                                      trie.struct_ref() if trie is not None else None,
                                      ptr(ids), ptr(lens), ptr(scores), ptr(ws), ws.numel(), stream_ptr()),
          "gdr_beam_search_table")
    return ids, lens, scores


def finish_generate_output(ids, lens, scores, max_length):
    """Host tail of generation_utils.py:905-919: width = min(max(len)+1, max_length); scores as Python floats."""
    sent_max_len = min(int(lens.max().item()) + 1, max_length)
    out = ids[:, :sent_max_len].contiguous(), scores.cpu().tolist()
    _ffi.check_device_fault("generate")                      # the host has just synchronised: a faulted launch must not pass
    return out
