"""SURVEY.md §5 debug aid: AddressSanitizer builds of the CPU-side code (`make -C oracle asan`, `make -C vm_asr_amd/csrc asan`),
exercised in child processes with the sanitizer runtime preloaded (CPU only: GPU ASan / xnack+ code objects are not available).

  * the oracle's C restatement runs its golden-vector tests (tests/test_oracle.py) under ASan;
  * the HIP library's HOST side (argument checks, geometry, slot / problem tables, workspace sizes — everything that runs before
    a launch) is driven through every entry point that needs no GPU, including the error paths, under ASan.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, args, timeout=900):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", **env_extra)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_oracle_goldens_under_address_sanitizer():
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt):
        pytest.skip("gcc has no libasan")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    r = _run({"LD_PRELOAD": rt, "VMASR_ORACLE_LIB": os.path.join(ROOT, "oracle", "libvmasr_oracle_asan.so"), "OMP_NUM_THREADS": "4"},
             ["-m", "pytest", "tests/test_oracle.py", "-x", "-q", "-p", "no:cacheprovider"])
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr + r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


_HOST_SCRIPT = r'''
import ctypes, os
from vm_asr_amd import _lib
l = _lib.lib()
assert l.vmasr_abi_version() == 1
# capability / geometry queries
for args in ((128, 512, 5, 3), (512, 1024, 5, 3), (1024, 1024, 5, 1), (32, 128, 5, 3), (100, 128, 5, 3)):
    l.vmasr_conv_mfma_supported(*args)
for d, h in ((8, 32), (16, 64), (64, 256), (128, 512), (7, 28)):
    l.vmasr_mlp_supported(d, h); l.vmasr_outproj_supported(2 * d, d); l.vmasr_inproj_supported(d, 4 * d, 4096)
l.vmasr_ss2d_supported(1, 1, 32, 128, 128); l.vmasr_ss2d_deep_supported(1, 2, 64, 64, 64); l.vmasr_ss2d_glue_supported(32, 16384, 0)
l.vmasr_ss2d_deep_waves_per_row(64, 64); l.vmasr_ss2d_part_floats(4, 32, 128, 128); l.vmasr_xproj_supported(1, 2, 64)
l.vmasr_layer_norm_bwd_workspace(4096, 64); l.vmasr_layer_norm_bwd_blocks(4096, 64); l.vmasr_small_linear_supported(1, 8)
l.vmasr_small_linear_bwd_workspace(1 << 20, 1, 8); l.vmasr_stft_bwd_workspace(4, 122640, 1024, 240); l.vmasr_istft_workspace(4, 512, 1024, 240)
l.vmasr_ln_gate_bwd_workspace(4, 64, 64, 64); l.vmasr_ln_gate_pair_supported(32, 128, 128); l.vmasr_conv_post_supported(1024, 3)
l.vmasr_stft_loss_blocks(); l.vmasr_masked_l1_blocks(); l.vmasr_sn_dot_blocks(); l.vmasr_adamw_chunk(); l.vmasr_sscan_chunk()
for k in range(_lib.K_COUNT):
    assert l.vmasr_prof_name(k)
# error paths: the argument checks run (and fill the error string) before anything touches a device
sl = (_lib.CgSlot * 2)()
assert l.vmasr_conv_mfma_fwd(sl, 2, 128, 128, 5, 3, 2, 256, 1, None) != 0 and l.vmasr_last_error()
assert l.vmasr_conv_mfma_fwd(sl, 2, 100, 128, 5, 3, 2, 256, 1, None) != 0
assert l.vmasr_conv_mfma_dgrad(sl, 2, 128, 128, 5, 3, 2, 256, None) != 0
assert l.vmasr_conv_mfma_wgrad(sl, 2, 128, 128, 5, 3, 2, 4, None) != 0
assert l.vmasr_split_bf16(None, None, None, 16, None) != 0
assert l.vmasr_im2col_kx1(None, None, 1, 8, 4, 5, 3, 2, 0, 0, None) != 0
p = _lib.SScanParams()
assert l.vmasr_sscan_fwd(ctypes.byref(p), None) != 0
bp = _lib.SScanBwdParams()
l.vmasr_sscan_bwd_workspace(ctypes.byref(bp))
assert l.vmasr_sscan_bwd(ctypes.byref(bp), None) != 0
sp = _lib.SS2DParams()
assert l.vmasr_ss2d_fwd(ctypes.byref(sp), None) != 0
dp = _lib.SS2DDeepParams()
assert l.vmasr_ss2d_deep_fwd(ctypes.byref(dp), None) != 0
n, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
assert l.vmasr_prof_collect(0, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)) == 0 and n.value == 0
assert _lib.prof_collect_shapes("ss2d_bwd_apply") == []
print("HOST-SIDE-OK")
'''


def test_hip_library_host_side_under_address_sanitizer():
    try:
        rt = subprocess.check_output(["/opt/rocm/bin/hipcc", "--print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    except Exception:
        pytest.skip("no hipcc")
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("clang has no ASan runtime")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "vm_asr_amd", "csrc"), "asan", "-j", str(min(8, os.cpu_count() or 1))],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run({"LD_PRELOAD": rt, "VMASR_LIB": os.path.join(ROOT, "vm_asr_amd", "libvmasr_hip_asan.so")}, ["-c", _HOST_SCRIPT])
    assert r.returncode == 0 and "HOST-SIDE-OK" in r.stdout and "AddressSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-3000:])
