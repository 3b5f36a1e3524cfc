"""The library is built without packed-fp32 VALU instructions (csrc/Makefile NOPK): with them one kernel mis-summed beside a second HIP
stream (profiles/r06_determinism_hunt.md).  Host-side check that the flag is in the build and still does what it is there for."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SRC = """#include <hip/hip_runtime.h>
__global__ void k(float2 *a, const float2 *b) { int i = threadIdx.x; float2 x = a[i], y = b[i]; x.x = x.x * y.x + 1.f; x.y = x.y * y.y + 1.f; a[i] = x; }
"""


def _nopk_flags():
    mk = open(os.path.join(ROOT, "vm_asr_amd", "csrc", "Makefile")).read()
    m = re.search(r"^NOPK\s*\?=\s*(.+)$", mk, re.M)
    assert m, "csrc/Makefile lost its NOPK flags"
    assert "$(NOPK)" in mk.split("$(BUILD)/%.o:")[1], "the object rule no longer passes $(NOPK)"
    return m.group(1).split()


def test_makefile_disables_packed_fp32():
    flags = _nopk_flags()
    assert "-packed-fp32-ops" in flags and "-target-feature" in flags


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_flag_removes_packed_fp32_instructions(tmp_path):
    src = tmp_path / "t.hip"
    src.write_text(SRC)

    def isa(extra):
        out = tmp_path / ("a.s" if extra else "b.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", *extra, "-o", str(out), str(src)], check=True,
                       capture_output=True, timeout=300)
        return out.read_text()
    assert "v_pk_fma_f32" in isa([])                 # what the compiler does on its own (the test would be vacuous otherwise)
    assert not re.search(r"v_pk_(fma|add|mul)_f32", isa(_nopk_flags()))
