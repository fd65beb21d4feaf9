"""The linear GEMM's stream-K tail (gemm_f32.hip: whole-tile rounds, then K-step ranges with exact accumulator hand-off over
the last tiles) must give the SAME BITS as the whole-tile kernel for every tile count: tail of 1..2 rounds, no tail
(T a multiple of the resident workgroups), between one and two tiles per CU (the 256-workgroup launch), padded and ragged
(device-side row count) batches.  The choice between the two
kernels is a process-wide tuning knob (GDR_GEMM_STREAMK, read once), so each mode runs in its own process and the
digests of the encoder outputs are compared.  Reference semantics are those of the encoder tests (modeling_t5.py:685-821);
this file only pins that a scheduling choice can never change a result."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
cfg.num_layers = 2                                 # two blocks: every linear shape, a fraction of the time
sd = synth.make_state_dict(cfg, seed=77, with_decoder=False)
enc = ops.T5EncoderHandle(cfg, sd, dev)
out = {}
for B, L in [(512, 40), (416, 40), (608, 40), (1024, 40), (700, 33), (512, 48), (48, 40), (64, 40), (80, 40), (128, 40), (160, 40)]:
    ids, mask = synth.make_tokens(B, L=L, seed=5 + B, min_len=8)
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    h, p = enc.forward(it, mt)
    _, pr = enc.forward(it, mt, want_hidden=False, ragged=True, live_rows_hint=int(mask.sum()))
    hr, _ = enc.forward(it, mt, want_pooled=False, ragged=True)          # no row hint: the grid is sized for B*L
    torch.cuda.synchronize()
    dg = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
    out[f"{B}x{L}"] = [dg(h), dg(p), dg(pr), dg(hr)]
print("DIGESTS " + json.dumps(out))
"""


def _run(mode, mid="512"):
    env = dict(os.environ, GDR_GEMM_STREAMK=mode, GDR_GEMM_STREAMK_MID=mid)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS ")][-1]
    return json.loads(line[len("DIGESTS "):])


def test_streamk_tail_is_bit_identical_to_whole_tiles_for_every_tile_count():
    whole = _run("0", "0")   # never: whole-tile kernels only (GDR_GEMM_STREAMK_MID=0: also between one and two tiles per CU)
    always = _run("1")       # every launch with more than one round of tiles takes the tail kernel
    default = _run("6")      # the shipped rule
    assert whole.keys() == always.keys() == default.keys()
    for k in whole:
        assert always[k] == whole[k], f"stream-K tail changed bits at batch {k}"
        assert default[k] == whole[k], f"default rule changed bits at batch {k}"
    # padded and ragged pooled outputs agree with each other as well (the ragged contract, test_gpu_ragged.py)
    for k, (h, p, pr, hr) in whole.items():
        assert p == pr, k
