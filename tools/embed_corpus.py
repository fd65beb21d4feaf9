#!/usr/bin/env python3
"""Offline corpus embedding with the doc tower — the role of Data_process/NQ_dataset/bert/bert.py:28-83 and its
launcher bert_NQ.sh:5-12 (one process per GPU, `--partition_num/--idx`, no collectives).

Input: pre-tokenised passages `tokens.npz` (input_ids int64[N,L<=128], attention_mask) — the tokenizer is out of scope
(SURVEY §2.3).  Output: `<out>/doc_embed.<idx>.npy` fp32[rows,768]; concatenate the shards in idx order to obtain the
corpus matrix D that gdr_sim_topk / gdr_rerank_topk read.

    python tools/embed_corpus.py --tokens tokens.npz --weights doc_tower.pt --partition_num 8 --idx 3 --out shards/
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import synth                                   # noqa: E402
from gdr_amd.dist import shard_bounds                       # noqa: E402
from gdr_amd.modeling import EncoderModel                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", required=True)
    ap.add_argument("--weights", default="", help="doc-tower state_dict (.pt); synthetic weights if empty")
    ap.add_argument("--partition_num", type=int, default=1)
    ap.add_argument("--idx", type=int, default=0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--padded", action="store_true",
                    help="compute every position of every padded batch as the reference does (default: the ragged form — PAD rows "
                         "are not computed, the embeddings are bit-identical)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: the bf16 precision mode of the doc tower (bf16 linear operands, fp32 accumulate; ragged form only) — "
                         "for a corpus that is kept in bf16 anyway (config C5)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    z = np.load(a.tokens)
    ids, mask = z["input_ids"], z["attention_mask"]
    lo, hi = shard_bounds(ids.shape[0], a.partition_num, a.idx)        # bert.py:51-61 partitioning
    bc = synth.bert_config(False)
    sd = torch.load(a.weights, map_location="cpu") if a.weights else synth.make_bert_state_dict(bc)
    enc = EncoderModel.from_state_dict(bc, sd, torch.device(a.device), dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32,
                                       ragged=not a.padded)
    out = np.empty((hi - lo, bc["hidden_size"]), dtype=np.float32)
    for s in range(lo, hi, a.batch):
        e = min(hi, s + a.batch)
        p = enc(passage={"input_ids": torch.from_numpy(ids[s:e]).to(a.device),
                         "attention_mask": torch.from_numpy(mask[s:e]).to(a.device)})
        out[s - lo:e - lo] = p.cpu().numpy()
    os.makedirs(a.out, exist_ok=True)
    np.save(os.path.join(a.out, f"doc_embed.{a.idx}.npy"), out)
    print(f"rows [{lo},{hi}) -> {a.out}/doc_embed.{a.idx}.npy")


if __name__ == "__main__":
    main()
