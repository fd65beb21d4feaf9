"""TEST INFRASTRUCTURE (see oracle/__init__.py) — docid codec, trie and metrics restated.

  * encode_single_newid : GDR_model/main_models.py:297-319
  * decode_token        : GDR_model/main_models.py:322-346
  * dec_2d              : GDR_model/main_utils.py:70-76
  * recall / MRR100     : GDR_model/main_metrics.py:194-267 (non-trivia branch)
"""
import numpy as np


def encode_single_newid(seq, kary=30, position=True):
    out = []
    if kary:
        for i, c in enumerate(seq.split("-")):
            out.append(i * kary + int(c) + 2 if position else int(c) + 2)
    else:
        for i, c in enumerate(seq):
            out.append(i * 10 + int(c) + 2 if position else int(c) + 2)
    return out + [1]


def decode_token(seqs, output_vocab_size=30, kary=30, position=True):
    result = []
    for seq in seqs:
        try:
            eos_idx = seq.tolist().index(1)
            seq = seq[1:eos_idx]
        except ValueError:
            pass                                   # reference prints "no eos token found" and keeps seq whole
        offset = np.arange(len(seq)) * output_vocab_size + 2 if position else 2
        res = seq - offset
        result.append(("-" if kary else "").join(str(c) for c in res))
    return result


def dec_2d(dec, size):
    return [dec[i:i + size] for i in range(0, len(dec), size)]


def recall_from_rows(rows, recall_num):
    """rows: iterable of (query, pred_csv, gt_csv, rank) as in the res1 TSV (main.py:244-247)."""
    q_gt, q_pred = {}, {}
    prev_q = ""
    for query, pred, gt, _rank in rows:
        if query != prev_q:
            q_pred[query] = pred.split(",")
            prev_q = query
        if query in q_gt:
            if len(q_gt[query]) <= 100:
                q_gt[query].add(gt)
        else:
            q_gt[query] = set(gt.split(","))
    out = {}
    for i in recall_num:
        total = 0
        for q in q_pred:
            hit = 0
            for p in q_gt[q]:
                if p in q_pred[q][:int(i)]:
                    hit = 1
            total += hit
        out[int(i)] = total / len(q_pred)
    return out


def mrr100_from_rows(rows):
    tot, n = 0.0, 0
    for _query, pred, gt, _rank in rows:
        pl = pred.split(",")
        if gt in pl:
            tot += 1 / (pl.index(gt) + 1)
        n += 1
    return tot / n
