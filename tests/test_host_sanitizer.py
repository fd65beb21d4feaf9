"""The host side of libgdr_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer ("ASan/UBSan on host
C-ABI code in CPU mode"): `make -C gdr_amd/csrc asan` builds the same sources with -fsanitize=address,undefined for the
host and unsanitized device code (GPU sanitizers are not available on this pool); a CPU-only child process loads it with the
ASan runtime preloaded and walks the host code of every entry point — argument validation, *_workspace_bytes, workspace
carving of the encoder / decode / prefix-table drivers (up to their first launch, which fails without a GPU), the
relative-position bucket table and the cluster-key hash.  Any sanitizer report makes the child exit non-zero."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import REPO

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
from gdr_amd import _ffi
l = _ffi.lib()
assert "asan" in _ffi.LIB_PATH
E = lambda rc: rc in (_ffi.GDR_EINVAL, _ffi.GDR_ENOSPC, _ffi.GDR_EHIP)
FAKE = C.c_void_p(0x7f0000000000)            # a non-null "device" pointer: the host code must never dereference it
dims = _ffi.GdrT5Dims(32128, 768, 64, 3072, 12, 2, 32, 128, 1e-6)
# ---- sizes
for B, L in ((1, 1), (64, 40), (512, 40), (0, 40), (-1, 5)):
    l.gdr_t5_encoder_workspace_bytes(C.byref(dims), B, L)
    l.gdr_t5_encoder_ragged_workspace_bytes(C.byref(dims), B, L)
    l.gdr_t5_encoder_bf16_workspace_bytes(C.byref(dims), B, L)
for args in ((4, 100, 768, 10, 0), (512, 320000, 768, 100, 0), (64, 320000, 768, 100, 1), (0, 0, 0, 0, 0)):
    l.gdr_sim_topk_workspace_bytes(*args)
assert l.gdr_rerank_workspace_bytes(64, 120) >= 64 * 120 * 4 and l.gdr_rerank_workspace_bytes(0, 5) == 0
l.gdr_beam_search_table_workspace_bytes(3, 100, 10, 30)
# ---- host tables
buf = (C.c_int32 * (128 * 128))()
for bi in (0, 1):
    assert l.gdr_t5_relative_bucket_table(bi, 32, 128, 128, 128, buf) == 0
assert E(l.gdr_t5_relative_bucket_table(1, 1, 128, 4, 4, buf))
toks = (C.c_int32 * 9)(*range(2, 11))
h = {l.gdr_cluster_key_hash(toks, n) for n in range(10)}
assert len(h) == 10
# ---- argument validation of every compute entry point (no launch is reached)
assert E(l.gdr_linear_f32(None, 0, None, 0, None, 0, 4, 4, 4, 0, None, None, 0, None))
assert E(l.gdr_linear_f32(FAKE, 3, FAKE, 4, FAKE, 4, 4, 4, 4, 0, None, None, 0, None))            # lda % 4
assert E(l.gdr_linear_f32(FAKE, 4, FAKE, 4, FAKE, 4, 4, 4, 4, 99, None, None, 0, None))           # epilogue
assert E(l.gdr_linear_f32_splitk(FAKE, 4, FAKE, 4, FAKE, 4, 4, 4, 4, 3, None, None, 0, None, 0, None))  # bias epilogue, no bias
assert E(l.gdr_linear_bf16(FAKE, 4, FAKE, 8, FAKE, 8, 4, 8, 8, 0, None, None, 0, None))
assert E(l.gdr_sim_topk(None, 4, None, 100, 768, 10, 0, None, None, None, 0, None, 0, None))
assert E(l.gdr_sim_topk(FAKE, 4, FAKE, 100, 768, 200, 0, FAKE, FAKE, None, 0, FAKE, 1 << 20, None))  # k > N
assert E(l.gdr_sim_topk_bf16(FAKE, 4, FAKE, 100, 770, 10, 0, FAKE, FAKE, None, 0, FAKE, 1 << 20, None))
# the bf16 pre-filter (r05): sizes, and every refusal before a launch
assert l.gdr_sim_topk_prefilter_workspace_bytes(512, 320000, 768, 100) > l.gdr_sim_topk_workspace_bytes(512, 320000, 768, 100, 0)
assert l.gdr_sim_topk_prefilter_workspace_bytes(0, 10, 768, 1) == 0
pw = l.gdr_sim_topk_prefilter_workspace_bytes(4, 1000, 768, 10)
assert E(l.gdr_sim_topk_prefilter(None, 4, None, None, 1.0, 1000, 768, 10, 0, None, None, None, None, 0, None))
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, 1.0, 1000, 770, 10, 0, FAKE, FAKE, None, FAKE, pw, None))       # d % 8
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, 1.0, 1000, 768, 2000, 0, FAKE, FAKE, None, FAKE, pw, None))     # k > N
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, 0.0, 1000, 768, 10, 0, FAKE, FAKE, None, FAKE, pw, None))       # dnorm_max
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, float("inf"), 1000, 768, 10, 0, FAKE, FAKE, None, FAKE, pw, None))
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, 1.0, 1000, 768, 10, 0, FAKE, FAKE, None, FAKE, 64, None))        # ENOSPC
assert E(l.gdr_sim_topk_prefilter(FAKE, 4, FAKE, FAKE, 1.0, 1000, 768, 10, 0, FAKE, FAKE, None, FAKE, pw, None))        # no GPU: EHIP
assert E(l.gdr_row_norm2_max(None, 10, 768, None, None))
assert E(l.gdr_row_norm2_max(FAKE, 10, 770, FAKE, None))
assert l.gdr_launch_count() >= 0
assert E(l.gdr_cast_f32_bf16(None, None, 8, None))
assert E(l.gdr_topk_merge(None, None, 2, 2, 2, None, None, None))
assert E(l.gdr_topk_pack(None, None, None, 2, 2, None, None))
assert E(l.gdr_topk_merge_packed(FAKE, 0, 2, 2, FAKE, FAKE, None, None))
assert E(l.gdr_l2_normalize(None, None, 2, 8, 1e-12, None))
assert E(l.gdr_t5_layer_norm(None, None, None, 2, 8, 1e-6, None))
assert E(l.gdr_t5_layer_norm(FAKE, FAKE, FAKE, 2, 6, 1e-6, None))       # d % 4 != 0
al = (C.c_float * 2)(0.0, 1.0)
assert E(l.gdr_rerank_topk(FAKE, FAKE, 768, FAKE, FAKE, FAKE, 4, 10, FAKE, 2, 10, 0, FAKE, FAKE, 9000, 0, 0, 100, 0, FAKE, 1 << 20, None))
assert E(l.gdr_rerank_topk(FAKE, FAKE, 768, FAKE, FAKE, FAKE, 4, 10, FAKE, 2, 10, 0, FAKE, FAKE, 120, 0, 0, 100, 0, FAKE, 16, None))  # ENOSPC
assert E(l.gdr_rerank_topk_bf16(FAKE, FAKE, 768, FAKE, FAKE, FAKE, 4, 10, FAKE, 2, 10, 2, FAKE, FAKE, 120, 0, 0, 100, 0, FAKE, 1 << 20, None))
ci = _ffi.GdrClusterIndex(10, 4, 12, FAKE.value, FAKE.value, FAKE.value, FAKE.value, FAKE.value)   # table_size not a power of two
assert E(l.gdr_cluster_candidates(C.byref(ci), FAKE, 2, 3, 10, FAKE, FAKE, FAKE, 36, None))
assert E(l.gdr_rerank_wire_pack(FAKE, FAKE, FAKE, None, 4, 768, 10, 120, FAKE, None))
assert E(l.gdr_rerank_wire_pack(FAKE, FAKE, FAKE, FAKE, 4, 768, 0, 120, FAKE, None))
assert E(l.gdr_rerank_wire_unpack(FAKE, 0, 768, 10, 120, FAKE, FAKE, FAKE, FAKE, None))
assert E(l.gdr_rerank_positions_to_ids(FAKE, FAKE, 4, 0, 120, FAKE, None))
# the sticky device-fault word (hipHostMalloc may fail without a GPU: then nothing is ever pending)
l.gdr_device_fault_clear()
assert l.gdr_device_fault_pending() == 0
l.gdr_device_fault_inject_for_tests()
p1 = l.gdr_device_fault_pending()
assert p1 in (0, 1) and l.gdr_device_fault_pending() == p1          # reading does not clear it
l.gdr_device_fault_clear()
assert l.gdr_device_fault_pending() == 0
# ---- drivers: host pointer tables with fake device pointers; validation + workspace carving run, the first launch fails
layers = (_ffi.GdrT5EncLayer * 2)()
for ly in layers:
    for f, _t in ly._fields_:
        setattr(ly, f, FAKE.value)
ew = _ffi.GdrT5EncoderWeights(dims, FAKE.value, FAKE.value, FAKE.value, layers)
need = l.gdr_t5_encoder_ragged_workspace_bytes(C.byref(dims), 8, 16)
assert E(l.gdr_t5_encoder_forward(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, FAKE, 64, None))           # ENOSPC
assert E(l.gdr_t5_encoder_forward(C.byref(ew), FAKE, FAKE, 8, 200, FAKE, None, FAKE, need, None))         # L > 128
assert E(l.gdr_t5_encoder_forward(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, FAKE, need, None))          # no GPU: EHIP
assert E(l.gdr_t5_encoder_forward_ragged(C.byref(ew), FAKE, FAKE, 8, 16, None, FAKE, -1, FAKE, need, None))
assert E(l.gdr_t5_encoder_forward_bf16(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, FAKE, need, None))
assert E(l.gdr_t5_encoder_forward_ragged_bf16(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, -1, FAKE, need, None))
dl = (_ffi.GdrT5DecLayer * 2)()
for ly in dl:
    for f, _t in ly._fields_:
        setattr(ly, f, FAKE.value)
alr = (_ffi.GdrAdaptorLayer * 1)()
for f, _t in alr[0]._fields_:
    setattr(alr[0], f, FAKE.value)
ddims = _ffi.GdrT5Dims(302, 768, 64, 3072, 12, 2, 32, 128, 1e-6)
dw = _ffi.GdrT5DecoderWeights(ddims, 30, 10, 1, 8, 2048, 1e-5, FAKE.value, FAKE.value, FAKE.value, FAKE.value, dl, alr,
                              FAKE.value, FAKE.value)
gneed = l.gdr_t5_generate_workspace_bytes(C.byref(dw), 4, 16, 10, 10)
assert gneed > 0 and l.gdr_t5_generate_workspace_bytes(C.byref(dw), 0, 16, 10, 10) == 0
for fn in (l.gdr_t5_generate, l.gdr_t5_generate_bf16):
    assert E(fn(C.byref(dw), FAKE, FAKE, 4, 16, 10, 10, 0.8, 10, None, None, FAKE, FAKE, FAKE, None, None, FAKE, 64, None))       # ENOSPC
    assert E(fn(C.byref(dw), FAKE, FAKE, 4, 16, 300, 10, 0.8, 10, None, None, FAKE, FAKE, FAKE, None, None, FAKE, gneed, None))   # beams
    assert E(fn(C.byref(dw), FAKE, FAKE, 4, 16, 10, 10, 0.8, 10, None, None, FAKE, FAKE, FAKE, None, None, FAKE, gneed, None))    # EHIP
trie = _ffi.GdrTrie(FAKE.value, FAKE.value, 5, 7)                                    # V mismatch
assert E(l.gdr_t5_generate(C.byref(dw), FAKE, FAKE, 4, 16, 10, 10, 0.8, 10, C.byref(trie), None, FAKE, FAKE, FAKE, None, None, FAKE, gneed, None))
ptab = _ffi.GdrPrefixTable(FAKE.value, 40, 30, 31, FAKE.value, FAKE.value, 3)     # 3 complete levels need 1 + 30 + 900 nodes
assert E(l.gdr_t5_generate(C.byref(dw), FAKE, FAKE, 4, 16, 10, 10, 0.8, 10, None, C.byref(ptab), FAKE, FAKE, FAKE, None, None, FAKE, gneed, None))
assert b"complete levels" in l.gdr_last_error()
lo = (C.c_int32 * 3)(0, 1, 4)
tneed = l.gdr_t5_prefix_table_workspace_bytes(C.byref(dw), 3)
assert E(l.gdr_t5_prefix_table_build(C.byref(dw), 2, lo, FAKE, FAKE, FAKE, FAKE, FAKE, 8, None))
assert E(l.gdr_t5_prefix_table_build(C.byref(dw), 2, lo, FAKE, FAKE, FAKE, FAKE, FAKE, tneed, None))
assert E(l.gdr_t5_prefix_table_build_bf16(C.byref(dw), 40, lo, FAKE, FAKE, FAKE, FAKE, FAKE, tneed, None))
assert E(l.gdr_beam_search_table(FAKE, 2, 30, 10, 10, 0.8, 10, None, FAKE, FAKE, FAKE, FAKE, 8, None))
bl = (_ffi.GdrBertLayer * 1)()
for f, _t in bl[0]._fields_:
    setattr(bl[0], f, FAKE.value)
bw = _ffi.GdrBertWeights(30522, 768, 12, 3072, 1, 512, 2, 1e-12, FAKE.value, FAKE.value, FAKE.value, FAKE.value, FAKE.value, bl)
bneed = l.gdr_bert_encoder_workspace_bytes(C.byref(bw), 2, 16)
assert E(l.gdr_bert_encoder_forward(C.byref(bw), FAKE, FAKE, None, 2, 16, FAKE, FAKE, FAKE, bneed, None))
assert E(l.gdr_bert_encoder_forward(C.byref(bw), FAKE, FAKE, None, 2, 16, FAKE, FAKE, FAKE, 8, None))
# r06: the doc tower's ragged / bf16 / split forms, the split linears and encoder, the tile-form query
rneed = l.gdr_bert_encoder_ragged_workspace_bytes(C.byref(bw), 2, 16)
assert rneed > bneed and l.gdr_bert_encoder_ragged_workspace_bytes(C.byref(bw), 0, 16) == 0
for fn in (l.gdr_bert_encoder_forward_ragged, l.gdr_bert_encoder_forward_ragged_bf16, l.gdr_bert_encoder_forward_ragged_split):
    assert E(fn(C.byref(bw), FAKE, FAKE, None, 2, 16, FAKE, FAKE, -1, FAKE, 8, None))                     # ENOSPC
    assert E(fn(C.byref(bw), FAKE, None, None, 2, 16, FAKE, FAKE, -1, FAKE, rneed, None))                  # null mask
    assert E(fn(C.byref(bw), FAKE, FAKE, None, 2, 16, FAKE, FAKE, -1, FAKE, rneed, None))                  # EHIP (first launch) or EINVAL
assert l.gdr_split_row_elems(768, 6) == 2304 and l.gdr_split_row_elems(768, 2) == 1536 and l.gdr_split_row_elems(100, 6) == 320
assert l.gdr_split_row_elems(0, 6) == 0
assert E(l.gdr_split_f32_bf16x3(None, None, 4, 768, 2304, None)) and E(l.gdr_split_f32_bf16x3(FAKE, FAKE, 4, 768, 100, None))   # ld_out < 3 K
assert E(l.gdr_split_f32_f16x2(FAKE, FAKE, 4, 768, 1000, None))                                                               # ld_out < 2 K
assert E(l.gdr_linear_split_bf16(FAKE, 2304, FAKE, 2304, FAKE, 768, 4, 768, 768, 5, 0, None, None, 0, None))   # terms
assert E(l.gdr_linear_split_bf16(FAKE, 1000, FAKE, 2304, FAKE, 768, 4, 768, 768, 6, 0, None, None, 0, None))   # lda < 3 K
assert E(l.gdr_linear_split_bf16(FAKE, 2304, FAKE, 2304, FAKE, 768, 4, 768, 768, 6, 3, None, None, 0, None))   # bias epilogue, no bias
assert E(l.gdr_linear_split_bf16(FAKE, 192, FAKE, 192, FAKE, 768, 4, 768, 96, 2, 0, None, None, 0, None))      # fp16 x 2 needs K % 128 == 0
sneed = l.gdr_t5_encoder_split_workspace_bytes(C.byref(dims), 8, 16)
assert sneed > need and l.gdr_t5_encoder_split_workspace_bytes(C.byref(dims), 0, 16) == 0
assert E(l.gdr_t5_encoder_forward_ragged_split(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, -1, 5, FAKE, sneed, None))    # terms
assert E(l.gdr_t5_encoder_forward_ragged_split(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, -1, 2, FAKE, 64, None))       # ENOSPC
assert E(l.gdr_t5_encoder_forward_ragged_split(C.byref(ew), FAKE, FAKE, 8, 16, FAKE, None, -1, 2, FAKE, sneed, None))    # EHIP
assert l.gdr_linear_bf16_tile_form(20480, 768, 3072, 0) == 256 and l.gdr_linear_bf16_tile_form(100, 768, 768, 0) == 64
assert l.gdr_linear_bf16_tile_form(100, 768, 100, 0) == 0
assert l.gdr_prof_enable(8) in (0, _ffi.GDR_EHIP, _ffi.GDR_EINVAL)
n_, ms_, w_ = (C.c_int64 * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
l.gdr_prof_collect(n_, ms_, w_)
assert isinstance(l.gdr_last_error(), bytes)
print("SANITIZED_OK")
'''


def test_host_side_is_clean_under_asan_and_ubsan():
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no ASan runtime in this ROCm image")
    p = subprocess.run(["make", "-C", os.path.join(REPO, "gdr_amd", "csrc"), "asan", "-j8"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lib = os.path.join(REPO, "gdr_amd", "libgdr_hip_asan.so")
    env = dict(os.environ, LD_PRELOAD=rt[-1], GDR_HIP_LIB=lib, PYTHONPATH=REPO, GDR_FFI_NO_TORCH="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", CHILD, REPO], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
