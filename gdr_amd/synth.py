"""Seeded synthetic inputs: model weights, NQ-320k-shaped corpus, queries, token ids, docids.

No checkpoint, tokenizer model or dataset ships with the reference
(/root/reference/.MISSING_LARGE_BLOBS:1-7), so everything the parity tests and
bench.py run on is generated here with numpy's PCG64 (bit-identical on every
box).  Distributions follow BASELINE.md §2 / SURVEY.md §8(d); weight scales
follow the reference initialiser (GDR_model/transformers/modeling_t5.py:607-637)
and the torch defaults of the modules the reference instantiates
(modeling_t5.py:1241-1244).
"""
import math

import numpy as np
import torch

from .config import GDRConfig


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_state_dict(cfg: GDRConfig, seed: int = 1234, with_decoder: bool = True,
                    ln_jitter: float = 0.1):
    """state_dict of the reference's ``T5ForConditionalGeneration`` (keys of SURVEY Appendix C,
    without the Lightning ``model.`` prefix).  ``ln_jitter`` perturbs norm weights away from 1.0
    so that a kernel ignoring them fails parity."""
    g = np.random.Generator(np.random.PCG64(seed))
    d, dk, H, ff = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
    inner = H * dk
    sd = {}

    def normal(shape, std):
        return _t(g.standard_normal(shape, dtype=np.float32) * np.float32(std))

    def ln(n):
        return _t(1.0 + ln_jitter * g.standard_normal(n, dtype=np.float32))

    def attn(prefix, rel_bias):
        sd[prefix + ".q.weight"] = normal((inner, d), (d * dk) ** -0.5)
        sd[prefix + ".k.weight"] = normal((inner, d), d ** -0.5)
        sd[prefix + ".v.weight"] = normal((inner, d), d ** -0.5)
        sd[prefix + ".o.weight"] = normal((d, inner), inner ** -0.5)
        if rel_bias:
            sd[prefix + ".relative_attention_bias.weight"] = normal(
                (cfg.relative_attention_num_buckets, H), d ** -0.5)

    sd["shared.weight"] = normal((cfg.vocab_size, d), 1.0)
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}"
        attn(p + ".layer.0.SelfAttention", i == 0)
        sd[p + ".layer.0.layer_norm.weight"] = ln(d)
        sd[p + ".layer.1.DenseReluDense.wi.weight"] = normal((ff, d), d ** -0.5)
        sd[p + ".layer.1.DenseReluDense.wo.weight"] = normal((d, ff), ff ** -0.5)
        sd[p + ".layer.1.layer_norm.weight"] = ln(d)
    sd["encoder.final_layer_norm.weight"] = ln(d)
    if not with_decoder:
        return sd

    Vd = cfg.decode_vocab_size
    sd["decode_embeddings.weight"] = normal((Vd, d), 1.0)
    sd["decoder.embed_tokens.weight"] = sd["decode_embeddings.weight"]
    sd["lm_head.weight"] = sd["decode_embeddings.weight"]
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}"
        attn(p + ".layer.0.SelfAttention", i == 0)
        sd[p + ".layer.0.layer_norm.weight"] = ln(d)
        attn(p + ".layer.1.EncDecAttention", i == 0)
        sd[p + ".layer.1.layer_norm.weight"] = ln(d)
        sd[p + ".layer.2.DenseReluDense.wi.weight"] = normal((ff, d), d ** -0.5)
        sd[p + ".layer.2.DenseReluDense.wo.weight"] = normal((d, ff), ff ** -0.5)
        sd[p + ".layer.2.layer_norm.weight"] = ln(d)
    sd["decoder.final_layer_norm.weight"] = ln(d)

    aff = cfg.adaptor_ff

    def uniform(shape, bound):
        return _t((g.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound))

    for i in range(cfg.adaptor_layer_num):
        p = f"adaptor.layers.{i}"
        for a in ("self_attn", "multihead_attn"):
            sd[f"{p}.{a}.in_proj_weight"] = uniform((3 * d, d), math.sqrt(6.0 / (4 * d)))  # xavier_uniform
            sd[f"{p}.{a}.in_proj_bias"] = uniform((3 * d,), 0.02)
            sd[f"{p}.{a}.out_proj.weight"] = uniform((d, d), d ** -0.5)
            sd[f"{p}.{a}.out_proj.bias"] = uniform((d,), 0.02)
        sd[f"{p}.linear1.weight"] = uniform((aff, d), d ** -0.5)
        sd[f"{p}.linear1.bias"] = uniform((aff,), d ** -0.5)
        sd[f"{p}.linear2.weight"] = uniform((d, aff), aff ** -0.5)
        sd[f"{p}.linear2.bias"] = uniform((d,), aff ** -0.5)
        for n in ("norm1", "norm2", "norm3"):
            sd[f"{p}.{n}.weight"] = ln(d)
            sd[f"{p}.{n}.bias"] = uniform((d,), 0.05)
    sd["adaptor_embeddings"] = _t(g.random((1, 1, d), dtype=np.float32))
    sd["adaptor_linear.weight"] = uniform((d * Vd, d), d ** -0.5)
    return sd


# ----------------------------------------------------------------------------------------------
# corpus / queries
# ----------------------------------------------------------------------------------------------
def make_corpus(N: int, d: int = 768, cluster_size: int = 12, seed: int = 20240320,
                chunk: int = 65536, rows=None):
    """``D fp32[N,d]``: doc i belongs to cluster i // cluster_size; d = c + noise.
    rows=(lo, hi): only rows [lo, hi) are materialised — bit-identical to ``make_corpus(N, ...)[lo:hi]`` (a rank of an N-GPU
    run keeps its shard only: 1/world of the host memory).  The noise is drawn from ONE PCG64 stream whose position cannot be
    jumped to (the float32 ziggurat consumes a data-dependent number of raw draws per sample), so the chunks in front of `lo`
    are still drawn (into a scratch chunk that is dropped) — the draws behind `hi` are not."""
    lo_r, hi_r = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    if not 0 <= lo_r <= hi_r <= N:
        raise ValueError(f"make_corpus: rows={rows} outside [0, {N}]")
    g = np.random.Generator(np.random.PCG64(seed))
    n_clusters = (N + cluster_size - 1) // cluster_size
    s = np.float32(1.0 / math.sqrt(d))
    cent = g.standard_normal((n_clusters, d), dtype=np.float32) * (s * np.float32(0.8))
    D = np.empty((hi_r - lo_r, d), dtype=np.float32)
    scratch = None
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        if lo >= hi_r:
            break
        if hi <= lo_r:                                   # wholly in front of the shard: advance the stream only
            if scratch is None:
                scratch = np.empty((chunk, d), dtype=np.float32)
            g.standard_normal((hi - lo, d), dtype=np.float32, out=scratch[:hi - lo])
            continue
        if lo >= lo_r and hi <= hi_r:                    # wholly inside: drawn in place
            blk = D[lo - lo_r:hi - lo_r]
            g.standard_normal((hi - lo, d), dtype=np.float32, out=blk)
            blk *= s * np.float32(0.6)
            blk += cent[np.arange(lo, hi) // cluster_size]
            continue
        blk = g.standard_normal((hi - lo, d), dtype=np.float32) * (s * np.float32(0.6))
        blk += cent[np.arange(lo, hi) // cluster_size]
        a, b = max(lo, lo_r), min(hi, hi_r)
        D[a - lo_r:b - lo_r] = blk[a - lo:b - lo]
    return D


def make_queries(D: np.ndarray, B: int, seed: int = 7):
    """``q = (5·d_g/‖d_g‖ + N(0,I)) / 3`` for a uniformly drawn gold doc g.  Returns (Q, gold)."""
    g = np.random.Generator(np.random.PCG64(seed))
    N, d = D.shape
    gold = g.integers(0, N, size=B)
    dg = D[gold]
    dg = dg / np.linalg.norm(dg, axis=1, keepdims=True)
    Q = (np.float32(5.0) * dg + g.standard_normal((B, d), dtype=np.float32)) / np.float32(3.0)
    return Q.astype(np.float32), gold.astype(np.int64)


def make_gold(N: int, B: int, seed: int = 7):
    """The gold doc ids make_queries(D[N, d], B, seed) draws, without D (a rank that holds only a shard of the corpus)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return g.integers(0, N, size=B).astype(np.int64)


def make_tokens(B: int, L: int = 40, vocab_hi: int = 32100, seed: int = 11, min_len: int = 8):
    """Token ids ``int64[B,L]`` uniform in [2, vocab_hi), lengths uniform min_len..L, EOS(1) last, PAD(0) after."""
    g = np.random.Generator(np.random.PCG64(seed))
    ids = g.integers(2, vocab_hi, size=(B, L)).astype(np.int64)
    lens = g.integers(min(min_len, L), L + 1, size=B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids[np.arange(B), lens - 1] = 1
    ids *= mask
    return ids, mask


def cluster_digits(c: int, depth: int, V: int):
    out = []
    for _ in range(depth):
        out.append(c % V)
        c //= V
    return out[::-1]


def make_cluster_ids(N: int, cluster_size: int = 12, V: int = 30):
    """Hierarchical docids: cluster c -> base-V digit string "a-b-c" (SURVEY §8d).
    Returns (cluster_strings list[n_clusters], depth, offsets int32[n_clusters+1], members int32[N])."""
    n_clusters = (N + cluster_size - 1) // cluster_size
    depth = 1
    while V ** depth < n_clusters:
        depth += 1
    names = ["-".join(str(x) for x in cluster_digits(c, depth, V)) for c in range(n_clusters)]
    offsets = np.minimum(np.arange(n_clusters + 1, dtype=np.int64) * cluster_size, N).astype(np.int32)
    members = np.arange(N, dtype=np.int32)
    return names, depth, offsets, members


def make_logit_table(B: int, max_len: int, Vd: int, eos_boost: float, seed: int):
    """Teacher-forcing logit table ``T[b, pos, last_token, :]``: lets beam-search tests reach the
    EOS / early-done / eviction paths that random weights almost never trigger."""
    g = np.random.Generator(np.random.PCG64(seed))
    table = g.standard_normal((B, max_len, Vd, Vd)).astype(np.float32)
    table[..., 1] += np.float32(eos_boost) * np.linspace(-1, 1, max_len)[None, :, None].astype(np.float32)
    return table


BERT_PREFIX = "ctx_encoder.bert_model."


def bert_config(tiny=False):
    """DPRConfig defaults = bert-base-uncased (transformers/configuration_dpr.py:81-93), or a tiny test shape."""
    if tiny:
        return dict(vocab_size=96, hidden_size=128, num_heads=2, d_ff=256, num_layers=2, max_pos=160, type_vocab=2, eps=1e-12)
    return dict(vocab_size=30522, hidden_size=768, num_heads=12, d_ff=3072, num_layers=12, max_pos=512, type_vocab=2, eps=1e-12)


def make_bert_state_dict(bcfg, seed=4321):
    """state_dict of the reference's doc tower `DPRContextEncoder` (keys of SURVEY Appendix C); BERT init scale 0.02,
    with jittered LayerNorm weights / biases so that a kernel ignoring them fails parity."""
    g = np.random.Generator(np.random.PCG64(seed))
    d, ff = bcfg["hidden_size"], bcfg["d_ff"]
    sd = {}

    def normal(shape, std=0.02):
        return _t(g.standard_normal(shape, dtype=np.float32) * np.float32(std))

    e = BERT_PREFIX + "embeddings."
    sd[e + "word_embeddings.weight"] = normal((bcfg["vocab_size"], d), 0.5)
    sd[e + "position_embeddings.weight"] = normal((bcfg["max_pos"], d), 0.2)
    sd[e + "token_type_embeddings.weight"] = normal((bcfg["type_vocab"], d), 0.2)
    sd[e + "LayerNorm.weight"] = _t(1.0 + 0.1 * g.standard_normal(d, dtype=np.float32))
    sd[e + "LayerNorm.bias"] = normal((d,), 0.05)
    for i in range(bcfg["num_layers"]):
        p = f"{BERT_PREFIX}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sd[p + f"attention.self.{n}.weight"] = normal((d, d), d ** -0.5)
            sd[p + f"attention.self.{n}.bias"] = normal((d,), 0.05)
        sd[p + "attention.output.dense.weight"] = normal((d, d), d ** -0.5)
        sd[p + "attention.output.dense.bias"] = normal((d,), 0.05)
        sd[p + "attention.output.LayerNorm.weight"] = _t(1.0 + 0.1 * g.standard_normal(d, dtype=np.float32))
        sd[p + "attention.output.LayerNorm.bias"] = normal((d,), 0.05)
        sd[p + "intermediate.dense.weight"] = normal((ff, d), d ** -0.5)
        sd[p + "intermediate.dense.bias"] = normal((ff,), 0.05)
        sd[p + "output.dense.weight"] = normal((d, ff), ff ** -0.5)
        sd[p + "output.dense.bias"] = normal((d,), 0.05)
        sd[p + "output.LayerNorm.weight"] = _t(1.0 + 0.1 * g.standard_normal(d, dtype=np.float32))
        sd[p + "output.LayerNorm.bias"] = normal((d,), 0.05)
    return sd
