"""Entry point that keeps the reference's command line (GDR_model/main.py:260-448, infer.sh:10-15).

    python -m gdr_amd.main --mode eval --decode_embedding 2 --num_return_sequences 10 --kary 30 ...   (infer.sh flags)

`--n_gpu N` (N > 1) with `--mode eval` fans out over N GPUs of the node: the command starts N fresh rank processes of itself
(gdr_amd/launch.py; the reference's analogues are Lightning's DDP spawn behind `--n_gpu`, main.py:57-70, and the per-GPU
launch of Data_process/NQ_dataset/bert/bert_NQ.sh:5-12), every rank decodes its own share of the query batches, the corpus is
row-sharded over the ranks and stage 2 runs as dist.ShardedIndex.rerank_own (BASELINE config C5's layout) — the TSVs and
metrics equal the one-GPU run's.

`--mode eval` runs the inference hot path on the MI355X: T5 encoder -> docid beam decode -> (with doc embeddings)
in-cluster dense rerank, writes the res1 TSV `query\\tpred\\tgt\\trank` (main.py:244-247) and prints recall@k / MRR100
(main_metrics.py:194-267).  `--mode calculate` recomputes the metrics of an existing TSV.  `--mode train` is out of
scope and exits with a message.

Every reference flag is accepted (plus `--trivia`, which infer.sh passes although the reference parser lacks it,
SURVEY fact 6).  Because neither the authors' checkpoint, tokenizer nor TSVs ship with the reference
(.MISSING_LARGE_BLOBS), inputs come from `--data_npz` (pre-tokenised: source_ids, source_mask, gt) and
`--infer_ckpt` (a Lightning .ckpt / state_dict); without them `--synthetic 1` (default) builds the seeded synthetic
NQ-320k-shaped workload of BASELINE.md.
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = "gdr_amd"

from . import _ffi, codec, launch, synth        # noqa: E402
from .config import GDRConfig, unsupported_variant   # noqa: E402

# (flag, type, default[, choices]) — names, types and defaults of main.py:262-396
_FLAGS = [
    ("output_dir", str, None), ("model_name_or_path", str, "t5-"), ("tokenizer_name_or_path", str, "t5-"),
    ("encoder_name_or_path", str, "bert--uncased"), ("encoder_tokenizer_name_or_path", str, "bert--uncased"),
    ("freeze_encoder", int, 0), ("freeze_embeds", int, 0), ("weight_decay", float, 1e-4), ("adam_epsilon", float, 1e-8),
    ("warmup_steps", int, 0), ("num_train_epochs", int, 500), ("gradient_accumulation_steps", int, 1),
    ("resume_from_checkpoint", str, None), ("n_val", int, -1), ("n_train", int, -1), ("n_test", int, -1),
    ("early_stop_callback", int, 0), ("fp_16", int, 0), ("opt_level", str, "O1"), ("max_grad_norm", float, 1.0),
    ("seed", int, 42), ("pretrain_encoder", int, 1), ("limit_val_batches", float, 1.0), ("softmax", int, 0),
    ("aug", int, 0), ("accelerator", str, "ddp"), ("num_layers", int, 12), ("num_decoder_layers", int, 6),
    ("d_ff", int, 3072), ("d_model", int, 1024), ("num_heads", int, 12), ("num_cls", int, 1000),
    ("decode_embedding", int, 2), ("output_vocab_size", int, 30), ("hierarchic_decode", int, 0),
    ("tie_word_embedding", int, 0), ("tie_decode_embedding", int, 1), ("gen_method", str, "greedy"),
    ("length_penalty", float, 0.8), ("random_gen", int, 0), ("label_length_cutoff", int, 0),
    ("check_val_every_n_epoch", int, 1), ("val_check_interval", float, 1.0), ("test_set", str, "dev"),
    ("train_batch_size", int, 128), ("eval_batch_size", int, 4), ("max_input_length", int, 40),
    ("inf_max_input_length", int, 40), ("max_output_length", int, 7), ("doc_length", int, 16),
    ("contrastive_variant", str, ""), ("num_return_sequences", int, 100), ("n_gpu", int, 1), ("devices", int, 8),
    ("mode", str, "train", ["train", "eval", "calculate"]), ("query_type", str, "gtq_doc"),
    ("learning_rate", float, 2e-4), ("decoder_learning_rate", float, 1e-4), ("certain_epoch", int, None),
    ("given_ckpt", str, ""), ("infer_ckpt", str, ""), ("model_info", str, "base", ["small", "large", "base", "3b", "11b"]),
    ("encoder_info", str, "base", ["base", "large"]), ("id_class", str, "bert_k30_c30_1"),
    ("ckpt_monitor", str, "recall", ["recall", "train_loss"]), ("Rdrop", float, 0), ("dropout_rate", float, 0.1),
    ("Rdrop_only_decoder", int, 0), ("Rdrop_loss", str, "KL", ["KL", "L2"]), ("adaptor_decode", int, 1),
    ("adaptor_efficient", int, 1), ("adaptor_layer_num", int, 4), ("test1000", int, 0), ("position", int, 1),
    ("contrastive", int, 0), ("embedding_distillation", float, 0.0), ("weight_distillation", float, 0.0),
    ("hard_negative", int, 0), ("aug_query", int, 0), ("aug_query_type", str, "corrupted_query"),
    ("sample_neg_num", int, 0), ("query_tloss", int, 0), ("weight_tloss", int, 0), ("ranking_loss", int, 0),
    ("disc_loss", int, 0), ("input_dropout", int, 1), ("denoising", int, 0), ("multiple_decoder", int, 0),
    ("decoder_num", int, 1), ("loss_weight", int, 0), ("nq", int, 1), ("kary", int, 30), ("tree", int, 1),
    ("ckpt_info", str, "334314test"), ("nodes", int, 1), ("docnum", int, 190727),
    ("project_path", str, "your project path"), ("kmeans_model", str, "ar2"), ("scheduler", str, "linear"),
    ("wandb_pro_name", str, "GDR"), ("is_train_encoder", int, 1), ("tau", float, 0.05), ("encoder_max_len", int, 128),
    ("max_intraclass_num", int, 10), ("use_query_embed_encoder", int, 1), ("use_query_embed_decoder_special", int, 0),
    ("use_query_embed_decoder_avg", int, 0), ("intra_rate", float, 1.0), ("train_encoder_epoch", int, 51),
    ("stage2_train_batchsize", int, 2), ("stage2_eval_batchsize", int, 2), ("begin_val_epoch", int, 0),
    ("doc_encoder_learning_rate", float, 2e-4), ("train_num", int, 256), ("eval_num", int, -1),
    ("data_suffix", str, "_30_2.5"), ("loss_func", str, "tanh"), ("fusion_strategy", str, "concate"),
    ("neg_sample_strategy", str, "random"),
    # tolerated / added
    ("trivia", int, 0),                      # infer.sh:15 passes it; only main_metrics.py reads it
    ("synthetic", int, 1), ("data_npz", str, ""), ("doc_embed_npy", str, ""), ("corpus_rows", int, 320000),
    ("n_queries", int, 512), ("res1_save_path", str, ""), ("device", str, "cuda:0"),
    ("pipeline_depth", int, 2),              # eval batches in flight on the GPU (each on its own stream) while the host decodes
                                             #    / formats the previous one; 1 = strictly one after another
    ("dist_backend", str, "nccl"),           # torch.distributed backend of the `--n_gpu N` ranks: nccl (= RCCL over xGMI), or gloo —
                                             #    collectives staged through the host, which lets several ranks share ONE GPU (tests)
    ("prefix_table", int, 1),                # 1: build the device prefix table over the corpus' docid trie at load
    ("constrain_tree", int, 0),              # 1: apply the trie constraint of generation_utils_previous.py:714-729 (the
                                             #    shipped generate() ignores decode_tree even with --tree 1)
]
_SIZES = {"base": (12, 6, 3072, 768, 12, 64), "large": (24, 12, 4096, 1024, 16, 64), "small": (6, 3, 2048, 512, 8, 64)}


def parsers_parser(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for f in _FLAGS:
        kw = dict(type=f[1], default=f[2])
        if len(f) > 3:
            kw["choices"] = f[3]
        parser.add_argument("--" + f[0], **kw)
    parser.add_argument("--recall_num", type=int, nargs="+", default=[1, 5, 10, 20, 50, 100])
    parser.add_argument("--score_rate", type=float, nargs="+", default=[0, 0.5, 1, 1.5, 2, 2.5, 3])
    parser.add_argument("--expand", type=bool, default=True)
    a = parser.parse_args(argv)
    # the reference's post-processing (main.py:398-447)
    a.dataset_name = "Self_NQ_{}_{}{}".format(a.kmeans_model, str(a.docnum), a.data_suffix)
    a.tokenizer_name_or_path += a.model_info
    a.model_name_or_path += a.model_info
    k = a.encoder_tokenizer_name_or_path.split("-")
    k[1] = a.encoder_info
    a.encoder_tokenizer_name_or_path = a.encoder_name_or_path = "-".join(k)
    a.gradient_accumulation_steps = max(int(8 / a.n_gpu), 1)
    if a.mode == "train" and "doc" in a.query_type:          # main.py:412-415
        assert a.contrastive_variant == ""
        a.max_input_length = a.doc_length
    if a.model_info in _SIZES:
        a.num_layers, a.num_decoder_layers, a.d_ff, a.d_model, a.num_heads, a.d_kv = _SIZES[a.model_info]
    if a.test1000:
        a.n_val = a.n_train = a.n_test = 1000
    return a


def set_seed(seed):
    """main_utils.py:12-18."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def deal_spans(n, batch_size, world, rank):
    """How `--n_gpu N` deals the query batches: span i = [i*batch_size, min(n, (i+1)*batch_size)) goes to rank i % world.
    The sharded stage 2 is a fixed-size collective, so every rank runs the SAME number of steps: a rank that ran out of spans
    repeats the last span (its output is dropped).  Returns (all_spans, my_spans, my_real) with len(my_spans) == ceil(len(all_spans) / world)
    on every rank."""
    all_spans = [(lo, min(n, lo + batch_size)) for lo in range(0, n, batch_size)]
    n_steps = (len(all_spans) + world - 1) // world
    mine = [all_spans[min(i * world + rank, len(all_spans) - 1)] for i in range(n_steps)]
    real = [i * world + rank < len(all_spans) for i in range(n_steps)]
    return all_spans, mine, real


def reassemble_spans(per_rank, n_spans, world):
    """The inverse on rank 0: per_rank[r] = rank r's REAL step outputs in its own order -> the outputs in span order."""
    return [per_rank[i % world][i // world] for i in range(n_spans)]


def _load_inputs(args, cfg, shard=None):
    """Returns dict(source_ids, source_mask, gt_cluster list[str], gt_doc list[str], index, doc_embed (np or None)).
    shard=(world, rank): a rank of `--n_gpu N` with the synthetic corpus materialises only its own rows (doc_embed = those rows,
    doc_rows = (lo, hi, N)); files are loaded whole."""
    if args.data_npz:
        z = np.load(args.data_npz, allow_pickle=False)
        index = codec.ClusterIndex([str(x) for x in z["cluster_names"]], z["cluster_offsets"], z["cluster_members"])
        doc = np.load(args.doc_embed_npy) if args.doc_embed_npy else None
        return dict(source_ids=z["source_ids"], source_mask=z["source_mask"], gt_cluster=[str(x) for x in z["gt_cluster"]],
                    gt_doc=[str(x) for x in z["gt_doc"]], index=index, doc_embed=doc)
    N = args.corpus_rows
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=args.kary)
    rows = None
    if shard is not None and shard[0] > 1:
        from .dist import shard_bounds
        rows = shard_bounds(N, shard[0], shard[1], cluster_size=12)
    D = synth.make_corpus(N, cfg.d_model, rows=rows)
    ids, mask = synth.make_tokens(args.n_queries, L=args.max_input_length, seed=11)
    gold = synth.make_gold(N, args.n_queries)            # the ids make_queries(D, n) draws (it needs the whole D only for the vectors)
    return dict(source_ids=ids, source_mask=mask, gt_cluster=[names[int(g) // 12] for g in gold],
                gt_doc=[str(int(g)) for g in gold], index=codec.ClusterIndex(names, offsets, members), doc_embed=D,
                doc_rows=(rows[0], rows[1], N) if rows is not None else None)


def inference(args):
    """main.py:115-250 on the MI355X path: generate cluster ids per query, write res1, print recall / MRR."""
    from .modeling import GDRModel, GDRRetriever
    cfg = GDRConfig.from_args(args)
    if args.infer_ckpt:
        # the reference loads whatever path it is given and fails if it is absent (main.py:121-126); so do we — a typo
        # must not turn into metrics of random weights.  weights_only: a checkpoint is tensors, not code.
        if not os.path.exists(args.infer_ckpt):
            raise FileNotFoundError(f"--infer_ckpt {args.infer_ckpt!r} does not exist (pass --infer_ckpt '' to run the "
                                    "seeded synthetic weights)")
        sd = torch.load(args.infer_ckpt, map_location="cpu", weights_only=True)
    else:
        print("[gdr_amd] --infer_ckpt is empty: using seeded synthetic weights (no trained checkpoint ships with the "
              "reference)")
        sd = synth.make_state_dict(cfg, seed=1234)
    world, rank, sharded_index = 1, 0, None
    if launch.under_launcher():                              # a rank of `--n_gpu N` (or of torch.distributed.run typed by hand)
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())   # more ranks than GPUs: they share
        torch.cuda.set_device(local_rank)
        args.device = f"cuda:{local_rank}"
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(args.device))
        else:
            dist.init_process_group(args.dist_backend)
        world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(args.device)
    two_stage_wanted = bool(args.is_train_encoder)
    data = _load_inputs(args, cfg, shard=(world, rank) if two_stage_wanted else None)
    if args.constrain_tree and args.kary != args.output_vocab_size:
        raise SystemExit(f"--constrain_tree 1 needs --kary ({args.kary}) == --output_vocab_size "
                         f"({args.output_vocab_size}): the trie is indexed by the head's digit columns")
    need_trie = args.constrain_tree or (args.prefix_table and args.kary == args.output_vocab_size)
    trie = codec.Trie.from_docids(data["index"].names, args.kary) if need_trie else None
    model = GDRModel(cfg, sd, dev, trie=trie if args.constrain_tree else None,
                     prefix_trie=trie if args.prefix_table else None, ragged=True)
    R = args.num_return_sequences
    two_stage = bool(args.is_train_encoder) and data["doc_embed"] is not None
    retr = None
    if two_stage and world > 1:
        # config C5's layout: rank r keeps rows [lo, hi) of the corpus (whole clusters when they are contiguous row blocks,
        # as the synthetic corpus' are; any row split is correct — a candidate is scored by the rank that holds its row)
        from .dist import ShardedIndex, shard_bounds
        if data.get("doc_rows"):                                 # the synthetic corpus: only this rank's rows were generated
            lo_r, hi_r, _n = data["doc_rows"]
            shard = torch.from_numpy(np.ascontiguousarray(data["doc_embed"], dtype=np.float32)).to(dev)
        else:
            N_rows = data["doc_embed"].shape[0]
            lo_r, hi_r = shard_bounds(N_rows, world, rank, cluster_size=12 if not args.data_npz else 1)
            shard = torch.from_numpy(np.ascontiguousarray(data["doc_embed"][lo_r:hi_r], dtype=np.float32)).to(dev)
        sharded_index = ShardedIndex(shard, lo_r)
        retr = GDRRetriever(model, None, data["index"], args, sharded=sharded_index)
    elif two_stage:
        retr = GDRRetriever(model, torch.from_numpy(np.ascontiguousarray(data["doc_embed"], dtype=np.float32)).to(dev),
                            data["index"], args)
    n = data["source_ids"].shape[0] if args.n_test < 0 else min(args.n_test, data["source_ids"].shape[0])
    texts = data.get("texts") or ["q%d" % i for i in range(data["source_ids"].shape[0])]
    inf_result_cache, outputs = [], []
    # N ranks: span i goes to rank i % N.  The sharded stage 2 is a fixed-size collective, so every rank runs the same number
    # of steps with the same batch size: the last span is padded with its last query, and ranks that ran out of spans repeat
    # the last one — padding rows and repeated steps are dropped below
    all_spans, spans, real = deal_spans(n, args.eval_batch_size, world, rank)
    full = args.eval_batch_size if world > 1 else 0

    def take(arr, lo, hi):
        sel = list(range(lo, hi)) + [hi - 1] * max(0, full - (hi - lo))
        return [arr[j] for j in sel] if isinstance(arr, list) else arr[sel]

    def batches():
        for lo, hi in spans:
            yield {"source_ids": torch.from_numpy(take(data["source_ids"], lo, hi)).to(dev),
                   "source_mask": torch.from_numpy(take(data["source_mask"], lo, hi)).to(dev), "texts": take(texts, lo, hi),
                   "gt": take(data["gt_cluster"], lo, hi), "oldid": take(data["gt_doc"], lo, hi)}

    def trimmed(out, lo, hi):                                    # drop the padding rows of a short last batch
        m = hi - lo
        return {"inf_result_batch": out["inf_result_batch"][:m], "inf_result_batch_prob": out["inf_result_batch_prob"][:m * R],
                "inf_index_batch": out["inf_index_batch"][:m]}

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if two_stage:
        try:
            for i, out in enumerate(retr.validation_steps(batches(), depth=max(1, args.pipeline_depth))):
                if real[i]:
                    outputs.append(trimmed(out, *spans[i]))
        except _ffi.GdrError:
            # a device fault (a stream-K hand-off that timed out) is sticky until acknowledged: this entry point owns the clear —
            # the run is reported as failed, but a process that catches the error and goes on is not poisoned for good
            if _ffi.clear_device_fault():
                print("inference: a device fault was pending; it has been acknowledged (gdr_device_fault_clear) — the step it "
                      "belongs to is invalid and the run is aborted", file=sys.stderr)
            raise
    else:
        for (lo, hi), b, is_real in zip(spans, batches(), real):
            if not is_real:
                continue
            outs, _ = model.generate(b["source_ids"], attention_mask=b["source_mask"], use_cache=False,
                                     max_length=args.max_output_length, num_beams=R, length_penalty=args.length_penalty,
                                     num_return_sequences=R, early_stopping=False, decode_embedding=args.decode_embedding,
                                     decode_vocab_size=args.output_vocab_size * args.max_output_length + 2)
            dec = codec.dec_2d(codec.decode_token(args, outs.cpu().numpy()), R)
            for j, pred in enumerate(dec[:hi - lo]):             # main.py:227-238 (a padded short batch: its real rows only)
                inf_result_cache.append([texts[lo + j], ",".join(pred), data["gt_cluster"][lo + j], 1])
    torch.cuda.synchronize()
    t_model = time.perf_counter() - t0
    if world > 1 or launch.under_launcher():
        # rank 0 writes the TSVs: collect every rank's step outputs (plain Python rows) in span order
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((outputs, inf_result_cache), gathered, dst=0)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return None, None
        per_rank_out = [g[0] for g in gathered]
        if two_stage:
            outputs = reassemble_spans(per_rank_out, len(all_spans), world)
        else:
            per = [g[1] for g in gathered]                       # rows of span i: rank i % world, its (i // world)-th block
            inf_result_cache, cursor = [], [0] * world
            for i, (lo, hi) in enumerate(all_spans):
                r_ = i % world
                inf_result_cache.extend(per[r_][cursor[r_]:cursor[r_] + (hi - lo)])
                cursor[r_] += hi - lo
    if two_stage:
        inf_result_cache = [row for out in outputs for row in out["inf_result_batch"]]
    # main.py:243-247: sort by (query, rank), keep rank 1, write the TSV
    res1 = sorted((r for r in inf_result_cache if r[3] == 1), key=lambda r: (r[0], r[3]))
    os.makedirs(os.path.dirname(args.res1_save_path) or ".", exist_ok=True)
    codec.write_res1(args.res1_save_path, res1)
    print(f"[gdr_amd] {n} queries, beam {R}, {world} GPU(s): {n / max(t_model, 1e-9):.1f} queries/s (model time only)")
    recall_value = codec.recall(args)
    mrr_value = codec.MRR100(args)
    if two_stage:
        # what Lightning validation adds on top of main.py's stage-1 numbers: the doc-level lists per alpha
        # (validation_step_i) and their metrics (validation_epoch_end, main_models.py:1643-1908)
        path2 = args.res1_save_path + ".docs.tsv"
        with open(path2, "w") as f:                              # query \t alpha \t doc ids csv \t gold doc
            for out in outputs:
                for per_alpha in out["inf_index_batch"]:
                    for ai, rows in enumerate(per_alpha):
                        for q, pred, gt in rows:
                            f.write(f"{q}\t{args.score_rate[ai]}\t{pred}\t{gt}\n")
        eargs = argparse.Namespace(**vars(args))
        by_size = {}                                             # the last batch may be short
        for out in outputs:
            by_size.setdefault(len(out["inf_index_batch"]), []).append(out)
        if len(by_size) == 1:
            eargs.eval_batch_size = next(iter(by_size))
            logged = codec.validation_epoch_end(outputs, eargs, strict=False)
        else:                                                    # re-batch to size 1 so that one pass covers all rows
            flat = [{"inf_result_batch": [o["inf_result_batch"][b]], "inf_result_batch_prob": [],
                     "inf_index_batch": [o["inf_index_batch"][b]]} for o in outputs for b in range(len(o["inf_index_batch"]))]
            eargs.eval_batch_size = 1
            logged = codec.validation_epoch_end(flat, eargs, strict=False)
        print("stage 2 (in-cluster rerank) doc-level metrics per alpha:")
        for al in args.score_rate:
            print("  alpha=%g  recall@1 %.4f  recall@10 %.4f  recall@100 %.4f  MRR100 %.4f" % (
                al, logged[f"recall1_{al}"], logged[f"recall10_{al}"], logged[f"recall100_{al}"], logged[f"MRR100_{al}"]))
        inference.last_logged = logged
    return recall_value, mrr_value


def calculate(args):
    return codec.recall(args), codec.MRR100(args)


def main(argv=None):
    args = parsers_parser(argv)
    set_seed(args.seed)
    dir_path = os.path.dirname(os.path.realpath(__file__))
    args.logs_dir = dir_path + "/logs/"
    if not args.res1_save_path:
        args.res1_save_path = args.logs_dir + "res1_recall{}_{}_{}.tsv".format(
            args.num_return_sequences, time.strftime("%Y%m%d-%H%M%S"), args.ckpt_info)
    if args.mode == "train":
        raise SystemExit("gdr_amd implements GDR's inference hot path only; --mode train is out of scope (SURVEY §2.2)")
    if args.mode == "eval":
        args.recall_num = [1, 5, 10, 20, 50, 100]
        why = unsupported_variant(args)                      # before any rank is started: a variant the kernels lack
        if why:
            raise SystemExit("gdr_amd: " + why)
        if args.n_gpu > 1 and not launch.under_launcher():
            # one command, N GPUs: this process (which has not touched a GPU) starts the N ranks as children and relays
            # their output; rank 0 writes the TSVs and prints the metrics
            import json
            tail = list(sys.argv[1:] if argv is None else argv)
            if "--res1_save_path" not in tail:
                tail += ["--res1_save_path", args.res1_save_path]
            rc, text = launch.spawn_ranks(args.n_gpu, tail, module="gdr_amd.main")
            if rc:
                raise SystemExit(rc)
            res = [ln for ln in text.splitlines() if ln.startswith("GDR_RESULT ")]
            r = json.loads(res[-1][len("GDR_RESULT "):]) if res else {}
            return r.get("recall"), r.get("mrr100")
        rec, mrr = inference(args)
        if launch.under_launcher() and rec is not None:
            import json
            print("GDR_RESULT " + json.dumps({"recall": rec, "mrr100": mrr}))
        return rec, mrr
    return calculate(args)


if __name__ == "__main__":
    main()
