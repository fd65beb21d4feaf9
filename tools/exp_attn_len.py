"""Where does the packed attention kernel's time go?  Same 512 x 12 (sequence, head) workgroups with all sequences at
length 8, 16, 24, 40 and the C2 mix: if the time barely follows the length, the kernel is bound by its per-workgroup
latency chain (load -> LDS -> barrier -> MFMA -> store), not by the matrix work."""
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
for name, lens in (("all 8", 8), ("all 16", 16), ("all 24", 24), ("all 40", 40), ("C2 mix 8..40", None)):
    ids, mask = synth.make_tokens(512, L=40, seed=11)
    if lens is not None:
        mask[:] = 0
        mask[:, :lens] = 1
    it, mt = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    f = lambda: enc.forward(it, mt, want_hidden=False, ragged=True, live_rows_hint=int(mask.sum()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    _ffi.check(lib.gdr_prof_enable(4096), "prof")
    for _ in range(5): f()
    torch.cuda.synchronize()
    n_l, ms_l, w_l = (C.c_int64 * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
    _ffi.check(lib.gdr_prof_collect(n_l, ms_l, w_l), "collect")
    print(f"{name:14s} rows={int(mask.sum()):6d}  attention {ms_l[3]/max(n_l[3],1)*1e3:6.1f} us/launch ({n_l[3]} launches)  linear {ms_l[0]/max(n_l[0],1)*1e3:6.1f} us/launch")
