#!/usr/bin/env python3
"""Per-stage throughput of the hot path on one MI355X (not the headline bench; feeds DESIGN.md).
  C3: full GDR = encoder -> docid beam decode (beam R) -> in-cluster rerank, B queries per batch
  doc tower: passages/s at L=128 (the corpus-embedding producer)
Usage: python tools/bench_stages.py [--B 64] [--beams 10] [--reps 5]"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import codec, ops, synth                      # noqa: E402
from gdr_amd.config import GDRConfig                        # noqa: E402
from gdr_amd.modeling import GDRModel, GDRRetriever, EncoderModel   # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--beams", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--corpus", type=int, default=320000)
    ap.add_argument("--skip-doc-tower", action="store_true")
    ap.add_argument("--no-prefix-table", action="store_true", help="compute the adaptor / head for every beam row every step")
    ap.add_argument("--padded", action="store_true", help="padded encoder form (PAD rows computed)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234)
    N = a.corpus
    names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
    t0 = time.perf_counter()
    model = GDRModel(cfg, sd, dev, ragged=not a.padded,
                     prefix_trie=None if a.no_prefix_table else codec.Trie.from_docids(names, 30))
    torch.cuda.synchronize()
    # random weights decode full-length rows; name clusters in that decoded form so that rerank has candidates
    D = torch.from_numpy(synth.make_corpus(N, cfg.d_model)).to(dev)
    ids, mask = synth.make_tokens(a.B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    R = a.beams
    args = types.SimpleNamespace(num_return_sequences=R, output_vocab_size=30, max_output_length=10, length_penalty=0.8,
                                 kary=30, position=1, score_rate=[0, 0.5, 1, 1.5, 2, 2.5, 3], loss_func="tanh")
    out = {"B": a.B, "beams": R, "corpus": N, "prefix_table": not a.no_prefix_table, "ragged_encoder": not a.padded,
           "model_load_s": time.perf_counter() - t0}
    if model.prefix_table is not None:
        out["prefix_table_nodes"], out["prefix_table_gb"] = model.prefix_table.n_table, model.prefix_table.nbytes() / 1e9
    t = timed(lambda: model.enc.forward(ids, mask, ragged=not a.padded), a.reps)
    out["encoder_qps"] = a.B / t
    gen = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8,
                                 num_return_sequences=R, output_scores=True, output_encoder_embedding=True)
    t = timed(gen, a.reps)
    out["generate_qps"], out["generate_ms"] = a.B / t, t * 1e3
    (dec, _), _ = gen()
    strs = sorted({s for s in codec.decode_token(args, dec.cpu().numpy())})
    # give every decoded string a real 12-doc cluster (synthetic weights do not know the corpus' ids)
    look = codec.ClusterIndex(strs + names[len(strs):], offsets, members)
    retr = GDRRetriever(model, D, look, args)
    batch = {"source_ids": ids, "source_mask": mask}
    t = timed(lambda: retr.validation_step_i(batch), a.reps)
    out["c3_two_stage_qps"], out["c3_ms"] = a.B / t, t * 1e3
    if not a.skip_doc_tower:
        bc = synth.bert_config(False)
        enc = EncoderModel.from_state_dict(bc, synth.make_bert_state_dict(bc), dev)
        pids, pmask = synth.make_tokens(256, L=128, vocab_hi=bc["vocab_size"], seed=3, min_len=32)
        pids, pmask = torch.from_numpy(pids).to(dev), torch.from_numpy(pmask).to(dev)
        t = timed(lambda: enc(passage={"input_ids": pids, "attention_mask": pmask}), a.reps)
        out["doc_tower_passages_per_s"] = 256 / t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
