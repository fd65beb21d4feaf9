"""Scheduling choices of the decode path must not change what generate() returns: the cross-attention that sums the q
projection's split-K slabs itself (GDR_DECODE_SLAB_Q; with finished q rows the MFMA beam-row form runs), the fused
reduce + residual + norm launch (GDR_DECODE_FUSE_NORM) and step 0 on one row per query instead of all B*R identical beam rows
(GDR_DECODE_DEDUP0) may differ in fp32 summation order only; leaving the step loop once every query is done
(GDR_DECODE_EARLY_EXIT, generation_utils.py:836-838) changes nothing at all.  Each switch selects code that also runs by default for other
shapes or modes (the bf16 mode, wide layers), so none keeps a dead kernel alive.  The switches are read once per process, so
each setting runs in its own process.  Semantics of the path: generation_utils.py:656-921, modeling_t5.py:1584-1652."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
model = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev)
out = {}
for B, R in ((1, 100), (5, 10), (64, 10)):
    ids, mask = synth.make_tokens(B, L=40, seed=21 + B)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    (dec, scores), _ = model.generate(ids, attention_mask=mask, max_length=8, num_beams=R, length_penalty=0.8,
                                      num_return_sequences=R, output_scores=True)
    out[f"{B}x{R}"] = {"ids": dec.cpu().tolist(), "scores": [float(s) for s in scores]}
# trie-constrained: every beam ends two steps after its docid's last digit, i.e. the call is done before max_length
from gdr_amd import codec
names = synth.make_cluster_ids(30000, cluster_size=12, V=30)[0]
tmodel = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev, trie=codec.Trie.from_docids(names, 30))
ids, mask = synth.make_tokens(5, L=40, seed=77)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
for rep in range(3):
    (dec, scores), _ = tmodel.generate(ids, attention_mask=mask, max_length=10, num_beams=10, length_penalty=0.8,
                                       num_return_sequences=10, output_scores=True)
out["trie_5x10"] = {"ids": dec.cpu().tolist(), "scores": [float(s) for s in scores]}
from gdr_amd import _ffi
torch.cuda.synchronize()
out["early_exits"] = int(_ffi.lib().gdr_t5_generate_early_exits())
out["last_done_step"] = int(_ffi.lib().gdr_t5_generate_last_done_step())     # of the trie-constrained call just above
print("RESULT " + json.dumps(out))
"""


def _run(**env):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_decode_scheduling_switches_do_not_change_generate():
    base = _run()
    base.pop("early_exits")            # the host half races with the GPU by design: not asserted (the device half below is)
    done_step = base.pop("last_done_step")
    # the device half of `if all(done): break` is deterministic: beams constrained to the 30 000-doc corpus' trie (docids of
    # depth 3) are all finished — EOS forced — by step depth + 2, and the kernel that sees the last query finish records it
    assert 0 < done_step <= 3 + 2, done_step
    # GDR_DECODE_EARLY_EXIT=0 runs every step although all queries are done (generation_utils.py:836-838 breaks there): done
    # queries only pad, so nothing may change — and the default run must actually have skipped (device side) or left early
    full = _run(GDR_DECODE_EARLY_EXIT="0")
    assert full.pop("early_exits") == 0 and full.pop("last_done_step") == 0
    assert full == base, "leaving the step loop when every query is done must not change any output"
    # GDR_DECODE_DEDUP0=0 runs step 0 on all B*R identical beam rows instead of one row per query: the same numbers from
    # launches of another shape (other split-K factors), i.e. fp32 summation order only
    # GDR_DECODE_SLAB_Q=0 reduces the cross-attention q projection in a launch of its own; the attention then gets finished q
    # rows and (round 4) takes the MFMA beam-row form instead of the generic kernel that sums the slabs: same scores, another
    # summation order — like the two switches after it
    # GDR_DECODE_FUSED=7 (r06, decode_fused.hip; off by default — measured slower): the self-attention, cross-attention and feed-forward
    # sub-blocks of a decoder block as (row panel, head / d_ff chunk) workgroups + one slab reduction each — the same products in
    # another fixed summation order
    for env in (dict(GDR_DECODE_SLAB_Q="0"), dict(GDR_DECODE_FUSE_NORM="0"), dict(GDR_DECODE_DEDUP0="0"), dict(GDR_DECODE_FUSED="7")):
        other = _run(**env)
        other.pop("early_exits"), other.pop("last_done_step")
        for key in base:
            a, b = base[key], other[key]
            sa, sb = np.asarray(a["scores"]), np.asarray(b["scores"])
            live = sa > -1e7                                   # dead-beam fillers tie at -1e9 in the reference too
            np.testing.assert_allclose(sb[live], sa[live], rtol=1e-4, atol=1e-4, err_msg=f"{env} {key}")
            ia, ib = np.asarray(a["ids"]), np.asarray(b["ids"])
            same = (ia == ib).all(axis=1)
            # a hypothesis may only trade places with one whose score is within the tolerance
            for r in np.nonzero(~same & live)[0]:
                assert abs(sa[r] - sb[r]) <= 1e-4 * max(1.0, abs(sa[r])), f"{env} {key} row {r}"
            assert same[live].mean() > 0.97, f"{env} {key}: {same[live].mean():.3f} of the live hypotheses identical"


SIM_CHILD = r"""
import json, sys, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
D = torch.from_numpy(synth.make_corpus(200000, 768, seed=5)).to(dev)
out = {}
for B in (1, 7, 32):
    Qn, _ = synth.make_queries(D[:20000].cpu().numpy(), B, seed=40 + B)
    vals, idx, status = ops.sim_topk(torch.from_numpy(Qn).to(dev), D, 100, exact_on_overflow=False, return_status=True)
    out[str(B)] = {"idx": idx.cpu().tolist(), "val": [float(x) for x in vals.flatten().cpu()], "status": int(status.max())}
print("RESULT " + json.dumps(out))
"""


def test_latency_mode_filter_lists_local_full_and_direct_agree():
    """sim_stream_f32_kernel<2> collects a workgroup's survivors per query in LDS and appends them with one atomic per
    (workgroup, query).  The lists are an unordered set until the select pass sorts them by (score, id), so the result must be
    bit-identical whether every survivor goes through LDS (default), the LDS lists overflow and the rest is appended directly
    (16 entries per query: the expected load is above that), or LDS is not used at all (0)."""
    def run(v, **extra):
        r = subprocess.run([sys.executable, "-c", SIM_CHILD, ROOT], env=dict(os.environ, GDR_SIM_LOCAL_LIST=v, **extra),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    base = run("192")
    assert all(v["status"] == 0 for v in base.values())
    assert run("16") == base
    assert run("0") == base
    # r06: the threshold and select tails of these calls run SLICED (sim_sliced_select_kernel: a query's list cut into <= 16 slices, one
    # workgroup each, the last arriver of a query merges the slices' top-k keys); with one workgroup per query (the r05 kernels,
    # GDR_SIM_SLICED=0) the result must be the same bits — keys are distinct, a top-k set does not depend on who arrives last
    assert run("192", GDR_SIM_SLICED="0") == base


BF16_CHILD = r"""
import json, sys, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
names = synth.make_cluster_ids(30000, cluster_size=12, V=30)[0]
model = GDRModel(cfg, synth.make_state_dict(cfg, seed=1234), dev, ragged=True, prefix_trie=codec.Trie.from_docids(names, 30),
                 dtype=torch.bfloat16)
out = {}
for B, R in ((512, 10), (140, 30), (64, 10)):          # 5 120 / 4 200 beam rows take the fused head, 640 rows the plain one
    ids, mask = synth.make_tokens(B, L=40, seed=31 + B)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    (dec, scores), _ = model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8,
                                      num_return_sequences=R, output_scores=True)
    out[f"{B}x{R}"] = {"ids": dec.cpu().tolist(), "scores": [float(s) for s in scores]}
print("RESULT " + json.dumps(out))
"""


def test_bf16_head_dot_in_the_gemm_epilogue_equals_the_plain_form():
    """bf16 mode, >= 1 024 beam rows: the head GEMM's epilogue dots its tile with the row's hidden state
    (launch_linear_bf16_headdot; 12 partial sums per (row, vocabulary entry)) instead of writing rows x 31 x 768 floats for
    head_logits to read back.  Same products, another fp32 summation tree: against GDR_DECODE_FUSE_NORM=0 (every linear writes
    its output and the consumer runs as its own launch — in bf16 mode that switch governs this fusion only) the hypothesis
    scores agree to 1e-4 and a hypothesis may only trade places with one whose score is that close.  The 640-row shape runs
    the plain form either way: identical."""
    def run(**env):
        r = subprocess.run([sys.executable, "-c", BF16_CHILD, ROOT], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    fused, plain = run(), run(GDR_DECODE_FUSE_NORM="0")
    assert fused["64x10"] == plain["64x10"]
    moved = 0
    for key in ("512x10", "140x30"):
        sa, sb = np.asarray(plain[key]["scores"]), np.asarray(fused[key]["scores"])
        live = sa > -1e7
        np.testing.assert_allclose(sb[live], sa[live], rtol=1e-4, atol=1e-4, err_msg=key)
        ia, ib = np.asarray(plain[key]["ids"]), np.asarray(fused[key]["ids"])
        same = (ia == ib).all(axis=1)
        for r in np.nonzero(~same & live)[0]:
            assert abs(sa[r] - sb[r]) <= 1e-4 * max(1.0, abs(sa[r])), f"{key} row {r}"
        assert same[live].mean() > 0.97, f"{key}: {same[live].mean():.3f} of the live hypotheses identical"
        moved += int((sa != sb).sum())
    assert moved > 0, "the fused head did not run (scores bit-identical to the plain form at >= 1 024 rows)"
