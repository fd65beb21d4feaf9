"""The ragged encoder form (gdr_t5_encoder_forward_ragged): PAD rows are not computed, pooled-only calls carry just the
CLS rows through the last block — and every kept row is BIT-IDENTICAL to the padded form, which itself is pinned against
the reference goldens (test_gpu_parity.py).  Reference semantics: transformers/modeling_t5.py:685-821 (T5Stack.forward),
modeling_utils.py:213-273 (extended mask), main_models.py:102-109 (CLS pool)."""
import numpy as np
import pytest
import torch

from gdr_amd.config import GDRConfig
from gdr_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def base_enc(dev):
    from gdr_amd import ops
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
    return cfg, sd, ops.T5EncoderHandle(cfg, sd, dev)


def _kept(mask):
    """Rows the ragged form computes (gdr_hip.h): the leading ones of a non-empty prefix mask, else the whole row."""
    m = np.asarray(mask) != 0
    B, L = m.shape
    keep = np.ones((B, L), bool)
    for b in range(B):
        n = int(m[b].sum())
        if n > 0 and m[b, :n].all():
            keep[b, n:] = False
    return keep


@pytest.mark.parametrize("B,L", [(512, 40), (128, 40), (130, 33), (64, 128), (64, 40), (32, 40), (100, 24), (8, 40)])
def test_ragged_rows_bit_identical_to_padded_form(dev, base_enc, B, L):
    """C2's batch (512 x 40, lengths 8..40), C3's 64-query encoder pass and other packable shapes — above 4 096 token rows
    on the un-split kernel forms, below on the split-K / stream-K forms the padded forward picks for the same B*L — :
    pooled (both ragged call forms) and every kept hidden row equal the padded form bit for bit; dropped rows are zeros."""
    cfg, sd, enc = base_enc
    ids, mask = synth.make_tokens(B, L=L, seed=11 + B, min_len=min(8, L))
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    h0, p0 = enc.forward(it, mt)
    h1, p1 = enc.forward(it, mt, ragged=True, live_rows_hint=int(mask.sum()))
    hn, p2 = enc.forward(it, mt, ragged=True, want_hidden=False)
    assert hn is None
    assert torch.equal(p1, p0), "pooled (hidden + pooled form)"
    assert torch.equal(p2, p0), "pooled (pooled-only form: last block on the CLS rows)"
    keep = torch.from_numpy(_kept(mask)).to(dev)
    assert torch.equal(h1[keep], h0[keep])
    assert int((h1[~keep] != 0).sum().item()) == 0
    assert float(keep.float().mean()) < 0.9                   # the batch really is ragged
    h3, pn = enc.forward(it, mt, ragged=True, want_pooled=False)
    assert pn is None and torch.equal(h3, h1)


def test_ragged_keeps_whole_rows_for_non_prefix_masks(dev, base_enc):
    """Masks that are not right padding — all zeros (softmax becomes uniform over every key, modeling_utils.py:271-272),
    a hole, left padding, full length, length 1 — keep all their positions and their mask: results equal the padded form
    bit for bit, and the oracle on those rows."""
    from oracle import t5_ref
    cfg, sd, enc = base_enc
    B, L = 128, 40
    ids, mask = synth.make_tokens(B, L=L, seed=5, min_len=4)
    mask[3] = 0                                               # fully masked
    mask[7] = 1
    mask[7, 10:14] = 0                                        # hole
    mask[9] = 0
    mask[9, 25:] = 1                                          # left padding
    mask[11] = 1                                              # full length
    mask[13] = 0
    mask[13, 0] = 1                                           # a single token
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    h0, p0 = enc.forward(it, mt)
    h1, p1 = enc.forward(it, mt, ragged=True)
    _, p2 = enc.forward(it, mt, ragged=True, want_hidden=False)
    keep_np = _kept(mask)
    assert keep_np[[3, 7, 9, 11]].all() and keep_np[13].sum() == 1
    keep = torch.from_numpy(keep_np).to(dev)
    assert torch.equal(p1, p0) and torch.equal(p2, p0)
    assert torch.equal(h1[keep], h0[keep]) and int((h1[~keep] != 0).sum().item()) == 0
    rows = [3, 7, 9, 13]
    ref = t5_ref.encoder_forward(sd, cfg, torch.from_numpy(ids[rows]), torch.from_numpy(mask[rows]))
    got = h1[rows].cpu()
    k = torch.from_numpy(keep_np[rows])
    torch.testing.assert_close(got[k], ref[k], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("kind,B,L", [("tiny", 6, 12), ("base", 4, 40)])
def test_ragged_contract_holds_when_the_padded_form_runs_internally(dev, kind, B, L):
    """d_kv != 64 or a batch too small to fill the chip: the entry point runs the padded form and zeroes the rows the packed
    form would have dropped — same outputs either way."""
    from gdr_amd import ops
    cfg = GDRConfig.tiny() if kind == "tiny" else GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=3, with_decoder=False)
    enc = ops.T5EncoderHandle(cfg, sd, dev)
    ids, mask = synth.make_tokens(B, L=L, vocab_hi=min(cfg.vocab_size, 32100), seed=2, min_len=2)
    mask[1] = 0
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    h0, p0 = enc.forward(it, mt)
    h1, p1 = enc.forward(it, mt, ragged=True)
    _, p2 = enc.forward(it, mt, ragged=True, want_hidden=False)
    keep = torch.from_numpy(_kept(mask)).to(dev)
    assert torch.equal(p1, p0) and torch.equal(p2, p0)
    assert torch.equal(h1[keep], h0[keep]) and int((h1[~keep] != 0).sum().item()) == 0


def test_generate_with_ragged_encoder_is_unchanged(dev):
    """GDRModel(ragged=True): decoded ids and scores equal the default path exactly (cross-attention masks the PAD keys
    whose rows are no longer computed), and the returned CLS rows — what validation_step_i reads, main_models.py:1466 —
    are bit-identical.  128 queries so that the packed form really runs."""
    from gdr_amd.modeling import GDRModel
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234)
    B, R = 128, 4
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    kw = dict(attention_mask=mt, max_length=10, num_beams=R, length_penalty=0.8, num_return_sequences=R, output_scores=True,
              output_encoder_embedding=True)
    (d0, s0), e0 = GDRModel(cfg, sd, dev).generate(it, **kw)
    (d1, s1), e1 = GDRModel(cfg, sd, dev, ragged=True).generate(it, **kw)
    assert torch.equal(d0, d1) and s0 == s1
    assert torch.equal(e0.last_hidden_state[::R][:, 0], e1.last_hidden_state[::R][:, 0])


@pytest.mark.parametrize("B,L", [(512, 40), (128, 40)])
def test_ragged_bf16_mode_bit_identical_to_padded_bf16(dev, B, L):
    """gdr_t5_encoder_forward_ragged_bf16: the packed form of the C5 precision mode — same rounding points, rows independent,
    so pooled (both call forms) and every kept hidden row equal gdr_t5_encoder_forward_bf16 bit for bit."""
    from gdr_amd import ops
    cfg = GDRConfig.base()
    sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
    enc = ops.T5EncoderHandle(cfg, sd, dev, dtype=torch.bfloat16)
    ids, mask = synth.make_tokens(B, L=L, seed=11 + B)
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    h0, p0 = enc.forward(it, mt)
    h1, p1 = enc.forward(it, mt, ragged=True, live_rows_hint=int(mask.sum()))
    _, p2 = enc.forward(it, mt, ragged=True, want_hidden=False)
    keep = torch.from_numpy(_kept(mask)).to(dev)
    assert torch.equal(p1, p0) and torch.equal(p2, p0)
    assert torch.equal(h1[keep], h0[keep]) and int((h1[~keep] != 0).sum().item()) == 0
