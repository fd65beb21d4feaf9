"""Does the partial last row panel (12 308 rows = 96 panels + 20 rows) cost more than its 1 % of the tiles?  Same tile
counts (97 row panels), with and without a ragged edge."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdr_amd import ops, synth, _ffi
from gdr_amd.config import GDRConfig
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = GDRConfig.base()
sd = synth.make_state_dict(cfg, seed=1234, with_decoder=False)
enc = ops.T5EncoderHandle(cfg, sd, dev)
lib = _ffi.lib()
def run(name, lens):
    ids, _ = synth.make_tokens(512, L=40, seed=11)
    mask = np.zeros((512, 40), np.int64)
    for b, n in enumerate(lens): mask[b, :n] = 1
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    f = lambda: enc.forward(it, mt, want_hidden=False, ragged=True, live_rows_hint=int(mask.sum()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    _ffi.check(lib.gdr_prof_enable(4096), "prof")
    for _ in range(10): f()
    torch.cuda.synchronize()
    n_l, ms_l, w_l = (C.c_int64 * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
    _ffi.check(lib.gdr_prof_collect(n_l, ms_l, w_l), "collect")
    print(f"{name:34s} rows={int(mask.sum()):6d}  linear {ms_l[0]/max(n_l[0],1)*1e3:6.1f} us/launch  total linear {ms_l[0]/10:6.3f} ms/pass")
base = [24] * 512
run("12288 rows (96 panels, no edge)", base)
run("12308 rows (96 panels + 20 rows)", [25] * 20 + [24] * 492)
run("12416 rows (97 panels, no edge)", [25] * 128 + [24] * 384)
run("12350 rows (96 panels + 62 rows)", [25] * 62 + [24] * 450)
_, m = synth.make_tokens(512, L=40, seed=11)
run("C2 mix (12308 rows)", [int(x) for x in m.sum(1)])
