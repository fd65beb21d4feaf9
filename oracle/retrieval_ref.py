"""TEST INFRASTRUCTURE (see oracle/__init__.py) — CPU restatement of the retrieval arithmetic.

  * dense similarity + top-k : GDR_model/dense.py:53-54 / encoder.py:128-129 (`q @ p.T`),
    `Tensor.topk(k, largest=True, sorted=True)` as at main_models.py:1625
  * CLS pool / pooler        : main_models.py:102-109, dense.py:18-27,39,50
  * in-cluster rerank        : main_models.py:1434-1462,1574-1637
"""
import torch


def cls_pool(hidden, pooler_w=None, pooler_b=None, normalize=False):
    """h[:,0] (main_models.py:102-109; dense.py:39,50), optional Linear + L2-normalise (dense.py:18-27)."""
    rep = hidden[:, 0]
    if pooler_w is not None:
        rep = rep @ pooler_w.T
        if pooler_b is not None:
            rep = rep + pooler_b
    if normalize:
        rep = torch.nn.functional.normalize(rep, dim=-1)
    return rep


def compute_similarity(q_reps, p_reps):
    """dense.py:53-54."""
    return torch.matmul(q_reps, p_reps.transpose(0, 1))


def sim_topk(Q, D, k, block=None):
    """scores = Q @ D.T then per-row top-k (values desc, int64 indices).  ``block`` bounds the temp."""
    if block is None:
        return compute_similarity(Q, D).topk(k, dim=1, largest=True, sorted=True)
    vals, idxs = [], []
    for lo in range(0, Q.shape[0], block):
        v, i = compute_similarity(Q[lo:lo + block], D).topk(k, dim=1, largest=True, sorted=True)
        vals.append(v)
        idxs.append(i)
    return torch.cat(vals), torch.cat(idxs)


def rerank(q, doc_embed, members_per_query, cluster_num_per_query, beam_scores, alphas, k, func="tanh"):
    """main_models.py:1574-1637 for one batch, block-diagonal only (each query against its own candidates).

    q [B,d]; doc_embed [N,d]; members_per_query[b] = list[int] candidate doc ids (concat over the R decoded
    clusters, beam order); cluster_num_per_query[b] = list[int] segment lengths (len R);
    beam_scores [B,R] length-penalised hypothesis scores.  Returns out[b][a] = (values[k], doc_ids[k])."""
    f = torch.tanh if func == "tanh" else torch.sigmoid
    prob = torch.softmax(torch.tensor(beam_scores, dtype=torch.float32).view(len(members_per_query), -1), dim=-1)
    out = []
    for b, (mem, cnum) in enumerate(zip(members_per_query, cluster_num_per_query)):
        Dc = doc_embed[torch.tensor(mem, dtype=torch.long)]
        sim = f(torch.mul(q[b].unsqueeze(0), Dc).sum(-1))                 # :1582 (own slice, :1607-1611)
        per_alpha = []
        for alpha in alphas:
            s = sim.clone()
            off = 0
            for j, n in enumerate(cnum):                                  # :1622-1624
                s[off:off + n] = s[off:off + n] + alpha * prob[b][j]
                off += n
            vals, idx = s.topk(k, dim=0, largest=True, sorted=True)       # :1625
            per_alpha.append((vals, torch.tensor([mem[i] for i in idx.tolist()], dtype=torch.long)))
        out.append(per_alpha)
    return out
