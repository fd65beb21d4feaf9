#!/usr/bin/env python3
"""Experiment: two generate() calls in flight on two HIP streams (decode kernels fill a fraction of the chip each).
Prints batches/s sequential vs overlapped.  Two model instances = two workspaces."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, R = int(os.environ.get("B", 64)), int(os.environ.get("BEAMS", 10))
cfg = GDRConfig.base(); sd = synth.make_state_dict(cfg, seed=1234)
models = [GDRModel(cfg, sd, dev), GDRModel(cfg, sd, dev)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ids, mask = synth.make_tokens(B, L=40, seed=11); ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
def launch(i):
    m = models[i]
    with torch.cuda.stream(streams[i]):
        enc_h, _ = m.enc.forward(ids, mask, want_pooled=False)
        return m.dec.generate(enc_h, mask, R, 10, 0.8, R)
def finish(out, i):
    with torch.cuda.stream(streams[i]):          # the read-back must wait on the stream that produced it
        return ops.finish_generate_output(out[0], out[1], out[2], 10)
for i in (0, 1): finish(launch(i), i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for k in range(n):
    finish(launch(k & 1), k & 1)
torch.cuda.synchronize(); t_seq = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
pend = launch(0)
for k in range(1, n):
    nxt = launch(k & 1)
    finish(pend, (k - 1) & 1)
    pend = nxt
finish(pend, (n - 1) & 1)
torch.cuda.synchronize(); t_ovl = (time.perf_counter() - t0) / n
ref = finish(launch(0), 0); chk = finish(launch(1), 1)
print(f"B={B} R={R}: sequential {t_seq*1e3:.2f} ms/batch ({B/t_seq:.0f} q/s), two in flight {t_ovl*1e3:.2f} ms/batch ({B/t_ovl:.0f} q/s), same result: {torch.equal(ref[0], chk[0])}")
