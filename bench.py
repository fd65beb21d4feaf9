#!/usr/bin/env python3
"""bench.py — queries/sec of GDR's dense-retrieval hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic input:
    T5-base encoder forward on `batch` tokenised queries (int64[batch,40])  ->  CLS pool  ->
    fused Q·Dᵀ + top-100 over the resident 320 000 x 768 fp32 corpus.
N = 1 runs BASELINE config C2 (batch 512, whole corpus on one GPU).  N > 1 (one process per GPU, launched by
torch.distributed.run) runs C4's layout with weak scaling: every rank encodes its own 512 queries, the corpus
is row-sharded N ways, pooled queries are all-gathered, each rank searches its shard for all 512·N queries,
and ONE all-to-all of the per-shard (score,id)[B,k] lists hands every rank the lists of its own 512 queries, which
it merges (gdr_amd/dist.py; --replicated-merge: all-gather + merge of all queries on every rank).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the fp32 MFMA GEMM that serves every
encoder linear: algorithmic flops per launch / average launch duration, both measured live over the timed
region with hipEvent pairs recorded by the library on the launch stream (gdr_prof_*).  `cpu_baseline` is the
oracle ("port" of the reference's CPU path) timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA (no sparsity)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="queries per GPU per step")
    ap.add_argument("--corpus", type=int, default=320000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the reference's precision (the headline); bf16 = config C5's precision mode: bf16 linear "
                         "operands in the encoder and a bf16 corpus, fp32 accumulate (not comparable with the f32 line)")
    ap.add_argument("--replicated-merge", action="store_true",
                    help="N > 1: all-gather the per-shard lists and merge all queries on every rank (instead of the "
                         "all-to-all that hands each rank the lists of its own queries)")
    ap.add_argument("--encoder", choices=["ragged", "padded"], default="ragged",
                    help="ragged (default): PAD token rows are not computed and only the CLS rows go through the last "
                         "block (exact: pooled output bit-identical to the padded form); padded: every one of the "
                         "batch x 40 rows through all 48 linears, as the reference does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    return ap.parse_args()


def host_threads():
    """Threads the CPU baseline may really use: scheduler affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sd, cfg, ids, mask, D, k, budget_s=25.0):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload: the
    sample doubles until one pass costs >= 1/4 of the budget, then is timed (median of 3)."""
    from oracle import t5_ref, retrieval_ref
    torch.set_num_threads(host_threads())
    Dt = torch.from_numpy(D)

    def run(n):
        h = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(ids[:n]), torch.from_numpy(mask[:n]))
        return retrieval_ref.sim_topk(retrieval_ref.cls_pool(h), Dt, k)

    n = 8
    run(n)                                           # warm-up (thread pool, page-in)
    while True:
        t0 = time.perf_counter()
        run(n)
        t = time.perf_counter() - t0
        if t >= budget_s / 8 or n >= ids.shape[0]:
            break
        n = min(n * 2, ids.shape[0])
    ts = [t]
    for _ in range(2):
        t0 = time.perf_counter()
        run(n)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[1]
    return {"value": n / med, "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} of the step's queries through the whole path (encoder fp32 + Q.D^T top-{k} over "
                      f"all {D.shape[0]} docs), torch-CPU oracle, warm-up + median of 3 ({med:.2f} s each)"}


def recall_at(idx, gold, ks=(1, 10, 100)):
    idx = np.asarray(idx)
    return [float(np.mean([(gold[b] in idx[b, :k]) for b in range(idx.shape[0])])) * 100.0 for k in ks]


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        _t = torch.ones(1, device=dev)
        dist.all_reduce(_t)                  # communicator set-up (RCCL connects lazily) stays out of the timed region
        torch.cuda.synchronize()             # even with --warmup 0
    torch.set_grad_enabled(False)

    from gdr_amd import ops, synth, _ffi
    from gdr_amd.config import GDRConfig
    from gdr_amd.dist import ShardedIndex, shard_bounds

    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
    bf16 = a.dtype == "bf16"
    enc = ops.T5EncoderHandle(cfg, sd, dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    D = synth.make_corpus(a.corpus, cfg.d_model)
    lo, hi = shard_bounds(a.corpus, world, rank, cluster_size=12)
    D_dev = torch.from_numpy(D[lo:hi]).to(dev)
    if bf16:
        D_dev = ops.to_bf16(D_dev)
    index = ShardedIndex(D_dev, lo, exact=False)     # no host sync in the timed region; the status is checked after it
    ids_all, mask_all = synth.make_tokens(a.batch * world, L=40, seed=11)
    ids = torch.from_numpy(ids_all[rank * a.batch:(rank + 1) * a.batch]).to(dev)
    mask = torch.from_numpy(mask_all[rank * a.batch:(rank + 1) * a.batch]).to(dev)

    ragged = a.encoder == "ragged" and not bf16
    live_rows = int(mask_all[rank * a.batch:(rank + 1) * a.batch].sum())      # token rows that are not PAD (host-side metadata)

    def step():
        _, pooled = enc.forward(ids, mask, want_hidden=False, ragged=ragged, live_rows_hint=live_rows)
        q_all = index.gather_queries(pooled)
        if a.replicated_merge:
            return index.search(q_all, a.k, return_status=True)
        return index.search_own(q_all, a.k, return_status=True)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    lib = _ffi.lib()
    launches_per_step = 13 * cfg.num_layers + 8         # per layer: 4 linears x (main + tail + reduce) + attention
    _ffi.check(lib.gdr_prof_enable(launches_per_step * a.steps + 16), "gdr_prof_enable")
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    n_l, ms_l, w_l = (C.c_int64 * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
    _ffi.check(lib.gdr_prof_collect(n_l, ms_l, w_l), "gdr_prof_collect")
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    overflowed = int(out[2].sum().item())     # gdr_sim_topk status: rows whose candidate list overflowed (subset top-k)
    if overflowed:
        raise SystemExit(f"bench: {overflowed} queries overflowed their candidate lists — the step did not compute "
                         "the exact top-k (use exact_on_overflow=True for such data)")
    total_q = a.batch * world * a.steps
    ms_per_step = dt / a.steps * 1e3
    result = None
    if rank == 0:
        def cls(c):
            if n_l[c] == 0:
                return None
            avg_ms = ms_l[c] / n_l[c]
            avg_work = w_l[c] / n_l[c]
            return {"launches": int(n_l[c]), "avg_ms": avg_ms, "gflop_per_launch": avg_work / 1e9,
                    "tflops": avg_work / (avg_ms * 1e-3) / 1e12, "share_of_step": ms_l[c] / (dt * 1e3)}

        lin, smp, flt, att, red = cls(0), cls(1), cls(2), cls(3), cls(7)
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("linear_gemm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "queries/sec on NQ-320k (768-d)", "value": total_q / dt, "unit": "queries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("C2" if world == 1 else "C4-layout") +
                       f": t5-base encoder on {a.batch} queries/GPU (L=40) + fused Q.D^T top-{a.k} over a "
                       f"{a.corpus}x{cfg.d_model} {'bf16' if bf16 else 'fp32'} corpus" +
                       (" [C5 precision mode: bf16 linear operands, fp32 accumulate]" if bf16 else "") +
                       (f" [ragged encoder: the {live_rows} non-PAD token rows of {a.batch * 40} are computed, last block "
                        "on the CLS rows only; pooled output bit-identical to the padded form]" if ragged else
                        " [padded encoder: all batch x 40 rows]") + ("" if world == 1 else f" row-sharded {world} ways, "
                       "all-gather of queries, all-to-all of per-shard top-k, local merge"),
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world, "seq_len": 40,
                       "corpus_rows": a.corpus, "dim": cfg.d_model, "k": a.k, "corpus_resident_in_hbm": True,
                       "encoder_rows": "ragged" if ragged else "padded", "live_token_rows_per_gpu": live_rows},
            "roofline": {"bound": "mfma", "kernel": ("gdr::gemm_nt_bf16_glds_kernel (bf16 operands, LDS-DMA staging; every encoder linear)"
                                                     if bf16 else "gdr::gemm_nt_f32_persistent_kernel (every encoder linear)"),
                         "achieved": lin["tflops"], "peak": BF16_MFMA_PEAK_TFLOPS if bf16 else F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s",
                         "frac": lin["tflops"] / (BF16_MFMA_PEAK_TFLOPS if bf16 else F32_MFMA_PEAK_TFLOPS),
                         "traffic": None if bf16 else traffic,
                         "launches": lin["launches"], "avg_launch_ms": lin["avg_ms"],
                         "algorithmic_gflop_per_launch": lin["gflop_per_launch"], "share_of_step": lin["share_of_step"]},
            "kernels": {"sim_sample_gemm": smp, "sim_filter_gemm": flt, "attention": att, "splitk_reduce": red},
        }
        if flt:
            # similarity as a whole (both GEMM passes): flops and the corpus bytes it must stream once
            sim_ms = (ms_l[1] + ms_l[2]) / a.steps
            rows = hi - lo
            result["kernels"]["sim_total"] = {
                "ms_per_step": sim_ms, "tflops": (w_l[1] + w_l[2]) / a.steps / (sim_ms * 1e-3) / 1e12,
                "frac_of_mfma_peak": (w_l[1] + w_l[2]) / a.steps / (sim_ms * 1e-3) / 1e12 /
                                     (BF16_MFMA_PEAK_TFLOPS if bf16 else F32_MFMA_PEAK_TFLOPS),
                "corpus_stream_gbs": rows * cfg.d_model * (2 if bf16 else 4) / (sim_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": rows * cfg.d_model * (2 if bf16 else 4) / (sim_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # The CPU leg (rank 0, N = 1 only): the oracle timed on the host cores, and — the metric's second half —
        # Recall@{1,10,100} of the GPU top-k against the oracle's on the same synthetic queries.  Nothing outside this
        # leg touches oracle/.
        result["recall"] = None
        if world == 1 and not a.no_cpu_baseline:
            if not a.no_recall:
                from oracle import retrieval_ref
                Q, gold = synth.make_queries(D, a.batch)
                _, gi = ops.sim_topk(torch.from_numpy(Q).to(dev), D_dev, a.k)
                _, ci = retrieval_ref.sim_topk(torch.from_numpy(Q), torch.from_numpy(D), a.k, block=128)
                result["recall"] = {"k": [1, 10, 100], "gpu": recall_at(gi.cpu().numpy(), gold),
                                    "cpu_oracle": recall_at(ci.numpy(), gold),
                                    "topk_ids_identical_rows": int((gi.cpu().numpy() == ci.numpy()).all(axis=1).sum()),
                                    "rows": a.batch}
            result["cpu_baseline"] = cpu_baseline(sd, cfg, ids_all, mask_all, D, a.k)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
