"""A/B of the sliced threshold / select tails of the latency-mode similarity (GDR_SIM_SLICED, sim_topk.hip): each setting in its own
process; prints ms per call at B = 1, 8, 32 (fp32 and bf16 corpus) and checks that the outputs are identical between the settings."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys, time, torch
sys.path.insert(0, sys.argv[1])
from gdr_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N, k = 320000, 100
Dn = synth.make_corpus(N, 768)
D = torch.from_numpy(Dn).to(dev)
Db = ops.to_bf16(D)
out = {}
for B in (1, 8, 32):
    Qn, _ = synth.make_queries(Dn[:50000], B, seed=3)
    Q = torch.from_numpy(Qn).to(dev)
    for name, Dm in (("f32", D), ("bf16", Db)):
        ws = ops.Workspace(dev)
        for _ in range(5): v, i, st = ops.sim_topk(Q, Dm, k, workspace=ws, exact_on_overflow=False, return_status=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): ops.sim_topk(Q, Dm, k, workspace=ws, exact_on_overflow=False)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        out[f"{name}_B{B}"] = {"ms": ms, "idx": i.cpu().tolist(), "val": [float(x) for x in v.flatten().cpu()], "status": int(st.max())}
print("RESULT " + json.dumps(out))
"""
res = {}
for v in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, GDR_SIM_SLICED=v), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res[v] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
for key in res["0"]:
    a, b = res["0"][key], res["1"][key]
    same = a["idx"] == b["idx"] and a["val"] == b["val"] and a["status"] == b["status"] == 0
    bytes_ = 320000 * 768 * (4 if key.startswith("f32") else 2)
    print(f"{key:9s} one-wg-per-query {a['ms']:.4f} ms   sliced {b['ms']:.4f} ms  ({bytes_ / b['ms'] / 1e6 / 8000:.3f} of HBM peak)   identical outputs: {same}")
