"""A/B of the fused decode sub-blocks (decode_fused.hip, GDR_DECODE_FUSED bit mask): generate() outputs against the unfused chain
(mask 0) and decode time / launches per call, each setting in its own process (the knob is read once).
usage: python tools/exp_fused_decode.py [masks, default 0,1,2,4,7]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys, time, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import synth, codec, _ffi
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
names = synth.make_cluster_ids(320000, cluster_size=12, V=30)[0]
trie = codec.Trie.from_docids(names, 30)
sd = synth.make_state_dict(cfg, seed=1234)
out = {}
for tab in (True, False):
    model = GDRModel(cfg, sd, dev, prefix_trie=trie if tab else None, ragged=True)
    for B, R in ((64, 10), (1, 100), (16, 10), (5, 30)):
        ids, mask = synth.make_tokens(B, L=40, seed=11)
        ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        g = lambda: model.generate(ids, attention_mask=mask, max_length=10, num_beams=R, length_penalty=0.8,
                                   num_return_sequences=R, output_scores=True)
        (dec, scores), _ = g()
        torch.cuda.synchronize()
        n0 = _ffi.lib().gdr_launch_count()
        g()
        torch.cuda.synchronize()
        launches = _ffi.lib().gdr_launch_count() - n0
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        te = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); model.enc.forward(ids, mask, want_pooled=False, ragged=True); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        out[f"{'tab' if tab else 'plain'}_{B}x{R}"] = {"ids": dec.cpu().tolist(), "scores": [float(x) for x in scores], "ms": sorted(ts)[len(ts) // 2] * 1e3,
                             "enc_ms": sorted(te)[len(te) // 2] * 1e3, "launches": int(launches)}
print("RESULT " + json.dumps(out))
"""


def run(mask):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, GDR_DECODE_FUSED=str(mask)), capture_output=True, text=True,
                       timeout=1500)
    if r.returncode != 0:
        print(f"mask {mask}: FAILED\n{r.stderr[-3000:]}")
        return None
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])


def main():
    import numpy as np
    masks = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,4,7".split(","))]
    base = run(0)
    for m in masks:
        res = base if m == 0 else run(m)
        if res is None:
            continue
        for key, v in res.items():
            a = base[key]
            sa, sb = np.asarray(a["scores"]), np.asarray(v["scores"])
            live = sa > -1e7
            err = float(np.abs(sa[live] - sb[live]).max()) if live.any() else 0.0
            same = float((np.asarray(a["ids"]) == np.asarray(v["ids"])).all(axis=1)[live].mean())
            print(f"mask {m} {key:14s} generate {v['ms']:7.3f} ms  decode {v['ms'] - v['enc_ms']:7.3f} ms  launches {v['launches']:5d}  "
                  f"max |score diff| {err:.2e}  ids identical {same:.3f}")


if __name__ == "__main__":
    main()
