"""CPU-side checks of the C-ABI boundary: the library loads without a GPU, exports every
symbol include/vmasr_hip.h declares, the ctypes mirrors match the C structs, and the host
wrappers reject bad inputs with the reference's error type (RuntimeError)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vmasr_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmasr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vm_asr_amd import _lib
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vmasr_hip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.vmasr_abi_version() == 1
    assert lib.vmasr_sscan_chunk() == 256


def test_struct_layout_matches_header():
    """Compile a tiny C program against the header and compare sizeof/offsetof."""
    from vm_asr_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "vmasr_hip.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(vmasr_sscan_params), offsetof(vmasr_sscan_params, A_d_stride),
         offsetof(vmasr_sscan_params, A_ptr), offsetof(vmasr_sscan_params, x_ptr),
         sizeof(vmasr_sscan_bwd_params), offsetof(vmasr_sscan_bwd_params, ws_bytes),
         sizeof(vmasr_spectral_item), offsetof(vmasr_spectral_item, R), offsetof(vmasr_spectral_item, col_tile_start));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(vmasr_ss2d_params), offsetof(vmasr_ss2d_params, x), offsetof(vmasr_ss2d_params, dDs),
         sizeof(vmasr_ss2d_deep_params), offsetof(vmasr_ss2d_deep_params, x), offsetof(vmasr_ss2d_deep_params, g32),
         offsetof(vmasr_ss2d_deep_params, gpos));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    P, Q = _lib.SScanParams, _lib.SScanBwdParams
    # vmasr_spectral_item is written from numpy (vm_asr_amd/discriminator.py:SpectralBatch): 5 pointers + 4 int32
    want = [ctypes.sizeof(P), P.A_d_stride.offset, P.A_ptr.offset, P.x_ptr.offset, ctypes.sizeof(Q), Q.ws_bytes.offset,
            56, 40, 52]
    S, D = _lib.SS2DParams, _lib.SS2DDeepParams
    want += [ctypes.sizeof(S), S.x.offset, S.dDs.offset, ctypes.sizeof(D), D.x.offset, D.g32.offset, D.gpos.offset]
    assert got == want


def test_host_checks_raise_runtime_error():
    from vm_asr_amd import selective_scan as ss
    u = torch.randn(2, 8, 16)
    A = -torch.rand(8, 1)
    Bm = torch.randn(2, 4, 1, 16)
    with pytest.raises(RuntimeError):  # CPU tensors: there is no CPU path in the product
        ss.fwd(u, u.clone(), A, Bm, Bm.clone(), None, None, True, 1)
    from vm_asr_amd import csm, dwconv, stft
    with pytest.raises(RuntimeError):
        csm.cross_scan(torch.randn(1, 2, 4, 4))
    with pytest.raises(RuntimeError):
        dwconv.dwconv3x3_silu(torch.randn(1, 2, 4, 4), torch.randn(2, 1, 3, 3), None)
    with pytest.raises(RuntimeError):
        stft.wav2spectro(torch.randn(1, 1, 2048), 128, 32, 128, "log2")


def test_invalid_params_rejected_by_c_abi():
    """Contract violations return a negative code before anything is launched (no GPU needed)."""
    from vm_asr_amd import _lib
    lib = _lib.lib()
    p = _lib.SScanParams()
    assert lib.vmasr_sscan_fwd(ctypes.byref(p), None) == -1
    assert b"non-positive" in lib.vmasr_last_error()
    p.batch, p.dim, p.seqlen, p.dstate, p.n_groups, p.n_chunks = 1, 6, 10, 1, 4, 1
    assert lib.vmasr_sscan_fwd(ctypes.byref(p), None) == -1
    assert b"dividable" in lib.vmasr_last_error()
    assert lib.vmasr_stft(None, None, None, 1, 1000, 1000, 10, 1000, 1, 1, None) == -1
    assert lib.vmasr_cross_scan(None, None, 1, 1, 1, 1, 0, None) == -1


def test_conv_mfma_launch_capability_covers_the_launchers_bounds():
    """the dispatch predicate must know what the launchers refuse (slots per launch: dgrad 24 / (stride + 1), wgrad 8;
    32-bit row offsets), so that an MPD with more periods or a longer segment falls back instead of failing mid-backward."""
    from vm_asr_amd import _lib
    q = _lib.lib().vmasr_conv_mfma_supported_launch
    assert q(128, 512, 5, 3, 5, 400_000) == 1 and q(1024, 1024, 5, 1, 5, 100_000) == 1      # the shipped 5-period MPD
    assert q(128, 512, 5, 3, 6, 400_000) == 1 and q(128, 512, 5, 3, 7, 400_000) == 0        # 7 x (3 + 1) > 24 dgrad problems
    assert q(1024, 1024, 5, 1, 8, 100_000) == 1 and q(1024, 1024, 5, 1, 9, 100_000) == 0    # wgrad: 8 slots
    assert q(128, 512, 5, 3, 5, (1 << 31) // (5 * 512)) == 0 and q(128, 512, 5, 3, 5, (1 << 31) // (5 * 512) - 1) == 1
    assert q(100, 512, 5, 3, 5, 1000) == 0 and q(128, 512, 5, 3, 0, 1000) == 0
