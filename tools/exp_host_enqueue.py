"""Where does a generate() call's host time go?  For each shape: the wall time of the C call alone (everything enqueued, nothing
waited for) against the wall time to completion, for one call at a time and for two calls issued from two host threads on two
HIP streams (ctypes releases the GIL for the duration of the C call, so the two enqueue loops really run side by side).
If enqueue ~= total the call is bound by the host's launch rate; if enqueue << total by the GPU's dependent-kernel chain."""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import codec, synth
from gdr_amd.config import GDRConfig
from gdr_amd.modeling import GDRModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
N = 320000
sd = synth.make_state_dict(cfg, seed=1234)
names, depth, offsets, members = synth.make_cluster_ids(N, cluster_size=12, V=30)
trie = codec.Trie.from_docids(names, 30)
model = GDRModel(cfg, sd, dev, ragged=True, prefix_trie=trie)
out = {}
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "64x10,1x100,16x10").split(",")]
for B, R in shapes:
    ids, mask = synth.make_tokens(B, L=40, seed=11)
    ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    enc_h, _ = model.enc.forward(ids, mask, want_pooled=False, ragged=True)

    def call():
        return model.dec.generate(enc_h, mask, R, 10, 0.8, R, prefix_table=model.prefix_table)

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(9):
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
    e1, t1_ = sorted(enq)[4] * 1e3, sorted(tot)[4] * 1e3
    # two calls from two host threads, each on its own stream (own workspace: ops.Workspace is per stream)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def worker(k, n, res):
        with torch.cuda.stream(streams[k]):
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            res[k] = time.perf_counter() - t0

    for n_threads in (1, 2):
        res = {}
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(k, 4, res)) for k in range(n_threads)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        out.setdefault(f"B{B}_beam{R}", {})[f"threads{n_threads}_x4calls"] = {
            "enqueue_ms_per_call": t_enq * 1e3 / (4 * n_threads), "total_ms_per_call": t_all * 1e3 / (4 * n_threads)}
    out[f"B{B}_beam{R}"].update({"enqueue_ms": e1, "total_ms": t1_})
print(json.dumps(out, indent=1))
