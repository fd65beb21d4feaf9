"""TEST INFRASTRUCTURE (see oracle/__init__.py) — CPU restatement of the reference's modified T5.

Every function cites the reference lines it restates (paths relative to
/root/reference/GDR_model/transformers/).  Plain torch-CPU fp32 ops in the reference's order;
weights come in as a ``state_dict`` with the reference's key names.
Pinned by tests/golden/*.npz generated from the imported reference (tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- relative positions
def relative_position_bucket(relative_position, bidirectional=True, num_buckets=32, max_distance=128):
    """modeling_t5.py:242-288.  relative_position = memory_pos - query_pos (int64 tensor)."""
    ret = torch.zeros_like(relative_position)
    n = -relative_position
    if bidirectional:
        num_buckets //= 2
        ret = ret + (n < 0).to(torch.long) * num_buckets
        n = torch.abs(n)
    else:
        n = torch.max(n, torch.zeros_like(n))
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    return ret + torch.where(is_small, n, val_if_large)


def compute_bias(qlen, klen, table, bidirectional, num_buckets=32):
    """modeling_t5.py:290-314.  table: [num_buckets, H] -> [1, H, qlen, klen]."""
    ctx = torch.arange(qlen, dtype=torch.long)[:, None]
    mem = torch.arange(klen, dtype=torch.long)[None, :]
    bucket = relative_position_bucket(mem - ctx, bidirectional=bidirectional, num_buckets=num_buckets)
    return table[bucket].permute(2, 0, 1).unsqueeze(0)


# --------------------------------------------------------------------------- layers
def t5_layer_norm(x, w, eps=1e-6):
    """modeling_t5.py:164-171 (RMS norm: fp32 variance, no mean subtraction, no bias)."""
    variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    x = x / torch.sqrt(variance + eps)
    return w * x


GEMM_BF16 = False   # set by bf16_linears(): emulate the C5 precision mode (linear operands rounded to bf16, fp32 accumulate)


class bf16_linears:
    """Context manager: every T5 linear rounds its activation and weight operands to bf16 (round-to-nearest-even) and
    accumulates in fp32 — what gdr_t5_encoder_forward_bf16 computes.  Norms, softmax, QK^T / PV and the residual
    stream stay fp32."""

    def __enter__(self):
        global GEMM_BF16
        self.prev, GEMM_BF16 = GEMM_BF16, True

    def __exit__(self, *a):
        global GEMM_BF16
        GEMM_BF16 = self.prev


def _lin(x, w):
    if GEMM_BF16:
        x, w = x.to(torch.bfloat16).to(torch.float32), w.to(torch.bfloat16).to(torch.float32)
    return x @ w.T


def t5_attention(x, kv, sd, prefix, H, dk, position_bias, round_qkv=False):
    """modeling_t5.py:316-421 without cache.  No 1/sqrt(d) scaling; fp32 softmax.
    round_qkv: the bf16 precision mode's encoder emits q, k, v as bf16 when d_kv = 64 (gdr_hip.h)."""
    bs = x.shape[0]

    def shape(t):
        if round_qkv:
            t = t.to(torch.bfloat16).to(torch.float32)
        return t.view(bs, -1, H, dk).transpose(1, 2)

    q = shape(_lin(x, sd[prefix + ".q.weight"]))
    src = x if kv is None else kv
    k = shape(_lin(src, sd[prefix + ".k.weight"]))
    v = shape(_lin(src, sd[prefix + ".v.weight"]))
    scores = torch.matmul(q, k.transpose(3, 2))
    scores = scores + position_bias
    weights = F.softmax(scores.float(), dim=-1).type_as(scores)
    ctx = torch.matmul(weights, v).transpose(1, 2).contiguous().view(bs, -1, H * dk)
    return _lin(ctx, sd[prefix + ".o.weight"])


def t5_ff(x, sd, prefix):
    """modeling_t5.py:181-186: wo(relu(wi(x))), no bias."""
    h = F.relu(_lin(x, sd[prefix + ".wi.weight"]))
    return _lin(h, sd[prefix + ".wo.weight"])


def encoder_forward(sd, cfg, input_ids, attention_mask, return_bias=False):
    """T5Stack.forward, encoder (modeling_t5.py:685-821); dropout is identity in eval()."""
    H, dk, eps = cfg.num_heads, cfg.d_kv, cfg.layer_norm_epsilon
    h = sd["shared.weight"][input_ids]                                    # :725
    L = input_ids.shape[1]
    ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -1e9   # modeling_utils.py:271-272
    bias = compute_bias(L, L, sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                        bidirectional=True, num_buckets=cfg.relative_attention_num_buckets)
    position_bias = bias + ext                                            # :399-400
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}"
        nx = t5_layer_norm(h, sd[p + ".layer.0.layer_norm.weight"], eps)
        h = h + t5_attention(nx, None, sd, p + ".layer.0.SelfAttention", H, dk, position_bias,
                             round_qkv=GEMM_BF16 and dk == 64 and L <= 128)
        nx = t5_layer_norm(h, sd[p + ".layer.1.layer_norm.weight"], eps)
        h = h + t5_ff(nx, sd, p + ".layer.1.DenseReluDense")
    h = t5_layer_norm(h, sd["encoder.final_layer_norm.weight"], eps)      # :803
    return (h, position_bias) if return_bias else h


def decoder_forward(sd, cfg, dec_ids, enc_hidden, enc_mask):
    """T5Stack.forward, decoder, use_cache=False (modeling_t5.py:685-821, block :498-584).
    dec_ids int64[R,t]; enc_hidden [R,L,d]; enc_mask [R,L]."""
    H, dk, eps = cfg.num_heads, cfg.d_kv, cfg.layer_norm_epsilon
    R, t = dec_ids.shape
    L = enc_hidden.shape[1]
    h = sd["decode_embeddings.weight"][dec_ids]
    seq = torch.arange(t)
    causal = (seq[None, None, :].repeat(R, t, 1) <= seq[None, :, None]).to(torch.float32)
    ext = (1.0 - causal[:, None, :, :]) * -1e9                            # modeling_utils.py:236-272
    self_bias = compute_bias(t, t, sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"],
                             bidirectional=False, num_buckets=cfg.relative_attention_num_buckets) + ext
    enc_ext = (1.0 - enc_mask[:, None, None, :].to(torch.float32)) * -1e9  # modeling_utils.py:179-211
    cross_bias = compute_bias(t, L, sd["decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight"],
                              bidirectional=True, num_buckets=cfg.relative_attention_num_buckets) + enc_ext
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}"
        nx = t5_layer_norm(h, sd[p + ".layer.0.layer_norm.weight"], eps)
        h = h + t5_attention(nx, None, sd, p + ".layer.0.SelfAttention", H, dk, self_bias)
        nx = t5_layer_norm(h, sd[p + ".layer.1.layer_norm.weight"], eps)
        h = h + t5_attention(nx, enc_hidden, sd, p + ".layer.1.EncDecAttention", H, dk, cross_bias)
        nx = t5_layer_norm(h, sd[p + ".layer.2.layer_norm.weight"], eps)
        h = h + t5_ff(nx, sd, p + ".layer.2.DenseReluDense")
    return t5_layer_norm(h, sd["decoder.final_layer_norm.weight"], eps)


# --------------------------------------------------------------------------- adaptor (nn.TransformerDecoder)
def _mha(x_q, x_kv, sd, prefix, nhead, attn_mask=None):
    """torch.nn.MultiheadAttention forward (batch_first=False): x_q [T,R,d], x_kv [S,R,d]."""
    T, R, d = x_q.shape
    S = x_kv.shape[0]
    hd = d // nhead
    Wi, bi = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = _lin(x_q, Wi[:d]) + bi[:d]
    k = _lin(x_kv, Wi[d:2 * d]) + bi[d:2 * d]
    v = _lin(x_kv, Wi[2 * d:]) + bi[2 * d:]
    q = q.contiguous().view(T, R * nhead, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.contiguous().view(S, R * nhead, hd).transpose(0, 1)
    v = v.contiguous().view(S, R * nhead, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        s = s + attn_mask
    w = F.softmax(s, dim=-1)
    o = torch.bmm(w, v).transpose(0, 1).contiguous().view(T, R, d)
    return _lin(o, sd[prefix + ".out_proj.weight"]) + sd[prefix + ".out_proj.bias"]


def adaptor_forward(sd, cfg, dec_ids):
    """modeling_t5.py:1615-1633: decode_embeddings(ids) through the post-LN nn.TransformerDecoder
    (ReLU, eps 1e-5, nhead 8) with memory = adaptor_embeddings broadcast [1,R,d] and a causal tgt mask.
    Returns [R,t,d]."""
    R, t = dec_ids.shape
    d = cfg.d_model
    x = sd["decode_embeddings.weight"][dec_ids].transpose(0, 1)           # [t,R,d]
    mem = (sd["adaptor_embeddings"] + torch.zeros(R, 1, 1)).transpose(0, 1)   # [1,R,d]
    mask = torch.full((t, t), float("-inf")).triu(1)
    eps = cfg.adaptor_ln_eps
    for i in range(cfg.adaptor_layer_num):
        p = f"adaptor.layers.{i}"
        x = F.layer_norm(x + _mha(x, x, sd, p + ".self_attn", cfg.adaptor_nhead, mask), (d,),
                         sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
        x = F.layer_norm(x + _mha(x, mem, sd, p + ".multihead_attn", cfg.adaptor_nhead), (d,),
                         sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
        ff = _lin(F.relu(_lin(x, sd[p + ".linear1.weight"]) + sd[p + ".linear1.bias"]), sd[p + ".linear2.weight"]) \
            + sd[p + ".linear2.bias"]
        x = F.layer_norm(x + ff, (d,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], eps)
    return x.transpose(0, 1)


# --------------------------------------------------------------------------- head
def valid_columns(p, V):
    """Columns that keep their logit at decoder position p (modeling_t5.py:1553-1557): p*V+2..p*V+V+1 and EOS(1)."""
    return list(range(p * V + 2, p * V + V + 2)) + [1]


def positional_mask(t, Vd, V):
    """select_valid_embedding's additive mask (modeling_t5.py:1546-1571) for positions 0..t-1: [t,Vd]."""
    m = torch.full((t, Vd), -1e9)
    for p in range(t):
        m[p, valid_columns(p, V)] = 0.0
    return m


def head_full(sd, cfg, dec_hidden, adapt_out):
    """Reference formulation, all positions and all columns (modeling_t5.py:1575-1576,1634-1646)."""
    d, Vd = cfg.d_model, cfg.decode_vocab_size
    R, t, _ = dec_hidden.shape
    seq_out = dec_hidden * (d ** -0.5)
    A = (adapt_out @ sd["adaptor_linear.weight"].T).reshape(R, t, d, -1)
    W = A + sd["lm_head.weight"].T.unsqueeze(0).unsqueeze(0)
    logits = torch.matmul(seq_out.unsqueeze(-2), W).squeeze(-2)
    return logits + positional_mask(t, Vd, cfg.output_vocab_size)[None]


def head_last_restricted(sd, cfg, dec_hidden_last, adapt_last, p):
    """Same logits for the last position only and only its valid columns:
    dec_hidden_last, adapt_last: [R,d].  Returns full-width [R,Vd] with -1e9 elsewhere."""
    d, Vd = cfg.d_model, cfg.decode_vocab_size
    R = dec_hidden_last.shape[0]
    cols = valid_columns(p, cfg.output_vocab_size)
    Wl = sd["adaptor_linear.weight"].view(d, Vd, d)[:, cols, :]           # [i, c', k]
    if GEMM_BF16:                                                         # the head GEMM is a linear of the C5 mode too
        adapt_last, Wl = adapt_last.to(torch.bfloat16).float(), Wl.to(torch.bfloat16).float()
    A = torch.einsum("rk,ick->ric", adapt_last, Wl)                       # [R, d(i), 31]
    W = A + sd["lm_head.weight"][cols].T.unsqueeze(0)
    h = dec_hidden_last * (d ** -0.5)
    lg = torch.einsum("ri,ric->rc", h, W)
    out = torch.full((R, Vd), -1e9)
    out[:, cols] = lg
    return out


def decode_logits(sd, cfg, dec_ids, enc_hidden, enc_mask, restricted=False):
    """T5ForConditionalGeneration.forward decode branch (modeling_t5.py:1529-1646) -> logits[:, -1, :]."""
    h = decoder_forward(sd, cfg, dec_ids, enc_hidden, enc_mask)
    a = adaptor_forward(sd, cfg, dec_ids)
    if restricted:
        return head_last_restricted(sd, cfg, h[:, -1], a[:, -1], dec_ids.shape[1] - 1)
    return head_full(sd, cfg, h, a)[:, -1, :]
