#!/usr/bin/env python3
"""Convert the reference's on-disk artefacts into the flat, mmap-able files gdr_amd reads.

  doc_embedding.pkl   (main_models.py:180-187: pickle, indexable by doc id -> tensor [1,768] / [768])  -> doc_embed.npy fp32[N,d]
  indexmap*.pkl       (main_models.py:874-889: pickle dict cluster-id string -> list[int] doc ids)     -> clusters.npz
                                                                                   (cluster_names, cluster_offsets, cluster_members)
  Lightning .ckpt     (main.py:121-126: {'state_dict': {'model.*', 'encoder.model.*'}})                -> t5.pt / doc_tower.pt state_dicts

None of these files ship with the reference (.MISSING_LARGE_BLOBS); the converters are exercised on synthetic
pickles of the same structure in tests/test_host_logic.py.

    python tools/convert_artifacts.py --doc_embedding doc_embedding.pkl --indexmap indexmap_insert.pkl --ckpt x.ckpt --out outdir
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd.codec import ClusterIndex                     # noqa: E402
from gdr_amd.modeling import strip_lightning_prefix        # noqa: E402


def convert_doc_embedding(obj):
    """list / dict / tensor of per-doc embeddings -> fp32 [N, d]."""
    if torch.is_tensor(obj):
        arr = obj.detach().cpu().float().numpy()
    elif isinstance(obj, np.ndarray):
        arr = obj.astype(np.float32)
    else:
        keys = sorted(obj.keys()) if isinstance(obj, dict) else range(len(obj))
        rows = []
        for k in keys:
            v = obj[k]
            v = v.detach().cpu().float().numpy() if torch.is_tensor(v) else np.asarray(v, dtype=np.float32)
            rows.append(v.reshape(-1))
        arr = np.stack(rows)
    return np.ascontiguousarray(arr.reshape(arr.shape[0], -1), dtype=np.float32)


def convert_indexmap(id_mapping):
    idx = ClusterIndex.from_id_mapping({str(k): [int(x) for x in v] for k, v in id_mapping.items()})
    return dict(cluster_names=np.array(idx.names), cluster_offsets=idx.offsets, cluster_members=idx.members)


def split_checkpoint(ckpt):
    sd = ckpt.get("state_dict", ckpt)
    t5 = strip_lightning_prefix(sd)
    tower = {k[len("encoder.model."):]: v for k, v in sd.items() if k.startswith("encoder.model.")}
    return t5, tower


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--doc_embedding")
    ap.add_argument("--indexmap")
    ap.add_argument("--ckpt")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.doc_embedding:
        with open(a.doc_embedding, "rb") as f:
            np.save(os.path.join(a.out, "doc_embed.npy"), convert_doc_embedding(pickle.load(f)))
    if a.indexmap:
        with open(a.indexmap, "rb") as f:
            np.savez(os.path.join(a.out, "clusters.npz"), **convert_indexmap(pickle.load(f)))
    if a.ckpt:
        t5, tower = split_checkpoint(torch.load(a.ckpt, map_location="cpu"))
        torch.save(t5, os.path.join(a.out, "t5.pt"))
        if tower:
            torch.save(tower, os.path.join(a.out, "doc_tower.pt"))


if __name__ == "__main__":
    main()
