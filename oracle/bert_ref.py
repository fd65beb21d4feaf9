"""TEST INFRASTRUCTURE (see oracle/__init__.py) — CPU restatement of the reference's doc tower.

`EncoderModel.forward(passage=...)` (GDR_model/main_models.py:79-89) = `DPRContextEncoder(...).pooler_output`
= `DPREncoder` (transformers/modeling_dpr.py:146-191: BertModel, pooled = sequence_output[:,0,:], projection_dim 0)
over `BertModel` (transformers/modeling_bert.py: embeddings :164-208, self-attention :236-275 with scores / sqrt(dh)
and the additive mask of the modified helper modeling_utils.py:271-272 = (1-m)*-1e9, erf-GeLU activations.py:23,
post-LN blocks, eps 1e-12).  state_dict keys as in SURVEY Appendix C (prefix `ctx_encoder.bert_model.`).
"""
import math

import torch
import torch.nn.functional as F

P = "ctx_encoder.bert_model."


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def bert_forward(sd, bcfg, input_ids, attention_mask, token_type_ids=None, bf16=False):
    """Returns (sequence_output [B,L,d], pooled [B,d]).
    bf16=True emulates the rounding points of gdr_bert_encoder_forward_ragged_bf16 (the reference has no such mode: this is the
    build's own definition of it, "parity unpinned"): every linear rounds its activation and weight operands to bf16 (RNE) and
    accumulates in fp32; q, k, v are emitted as bf16 (q already scaled by 1/sqrt(dh), an exact power of two at dh = 64);
    embeddings, biases, LayerNorm, softmax and the residual stream stay fp32."""
    d, H, eps = bcfg["hidden_size"], bcfg["num_heads"], bcfg["eps"]
    dh = d // H
    B, L = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = P + "embeddings."
    x = sd[e + "word_embeddings.weight"][input_ids] + sd[e + "position_embeddings.weight"][torch.arange(L)][None] \
        + sd[e + "token_type_embeddings.weight"][token_type_ids]
    x = F.layer_norm(x, (d,), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -1e9

    def heads(t):
        return t.view(B, L, H, dh).permute(0, 2, 1, 3)

    def lin(a, w):
        return (_r16(a) @ _r16(w).T) if bf16 else a @ w.T

    sc = 1.0 / math.sqrt(dh)
    for i in range(bcfg["num_layers"]):
        p = f"{P}encoder.layer.{i}."
        if bf16:   # the scale is folded into the q rows of the weight and the bias before the weight is rounded
            q = heads(_r16(lin(x, sd[p + "attention.self.query.weight"] * sc) + sd[p + "attention.self.query.bias"] * sc))
            k = heads(_r16(lin(x, sd[p + "attention.self.key.weight"]) + sd[p + "attention.self.key.bias"]))
            v = heads(_r16(lin(x, sd[p + "attention.self.value.weight"]) + sd[p + "attention.self.value.bias"]))
            s = torch.matmul(q, k.transpose(-1, -2)) + ext
        else:
            q = heads(x @ sd[p + "attention.self.query.weight"].T + sd[p + "attention.self.query.bias"])
            k = heads(x @ sd[p + "attention.self.key.weight"].T + sd[p + "attention.self.key.bias"])
            v = heads(x @ sd[p + "attention.self.value.weight"].T + sd[p + "attention.self.value.bias"])
            s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + ext
        ctx = torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3).contiguous().view(B, L, d)
        t = lin(ctx, sd[p + "attention.output.dense.weight"]) + sd[p + "attention.output.dense.bias"]
        x = F.layer_norm(t + x, (d,), sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        f = gelu(lin(x, sd[p + "intermediate.dense.weight"]) + sd[p + "intermediate.dense.bias"])
        t = lin(f, sd[p + "output.dense.weight"]) + sd[p + "output.dense.bias"]
        x = F.layer_norm(t + x, (d,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x, x[:, 0, :]
