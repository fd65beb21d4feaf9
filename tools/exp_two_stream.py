"""Experiment: does running two half-batches of the encoder on two streams hide GEMM tails / launch ramps?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdr_amd import ops, synth
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
e1 = ops.T5EncoderHandle(cfg, sd, dev)
ids, mask = synth.make_tokens(512, L=40, seed=11)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
ws2 = ops.Workspace(dev)
def one():
    e1.forward(ids, mask)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("single stream  B=512: %.2f ms" % bench(one))
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    wss = [ops.Workspace(dev) for _ in range(parts)]
    chunk = 512 // parts
    def multi():
        cur = torch.cuda.current_stream()
        for p, (s, w) in enumerate(zip(streams, wss)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e1.ws = w
                e1.forward(ids[p * chunk:(p + 1) * chunk], mask[p * chunk:(p + 1) * chunk])
        for s in streams:
            cur.wait_stream(s)
    print("%d streams x B=%d: %.2f ms" % (parts, chunk, bench(multi)))
