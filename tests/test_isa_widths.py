"""Static check of the device code hipcc emits for gfx950 (no GPU needed): the two-rows-per-lane SpMV is only worth having if
its gathers really are 16-byte (fp64) / 8-byte (fp32) buffer loads and its stores 16-byte -- one compiler quirk met while it was
written turned the fp32 pair load into a 4-byte load that returned the first element twice (csrc/mik_sell.h, buffer_gather2)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "iterativesolvers.jl_amd", "csrc", "mik_core.hip")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC) and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "core.s")
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", SRC, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def body(text, mangled_prefix):
    m = re.search(r"^(%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(mangled_prefix), text, re.S | re.M)
    assert m, mangled_prefix + ": kernel not found in the device assembly"
    return m.group(2)


def count(b, op):
    return len(re.findall(r"^\s*%s\b" % re.escape(op), b, re.M))


def test_two_row_spmv_uses_wide_gathers(asm):
    # k_spmv_sdiab2<double, fused dot, nt, 7 slots, centre 3>: x[2l], x[2l+1] + the 4 gathered slots = 5 16-byte loads on the
    # class path, one 16-byte store, the 16-bit mask load, whole-wave DPP shifts for the lane neighbours
    b = body(asm, "_Z13k_spmv_sdiab2IdLb1ELb1ELi7ELi3EE")
    assert count(b, "buffer_load_dwordx4") >= 5 and count(b, "buffer_store_dwordx4") >= 1
    assert count(b, "buffer_load_ushort") == 1
    assert "wave_shr:1" in b and "wave_shl:1" in b
    # the fp32 kernel carries its two rows in 8-byte loads and stores
    f = body(asm, "_Z13k_spmv_sdiab2IfLb1ELb1ELi7ELi3EE")
    assert count(f, "buffer_load_dwordx2") >= 5 and count(f, "buffer_store_dwordx2") >= 1


def test_one_row_spmv_takes_lane_neighbours_by_dpp(asm):
    b = body(asm, "_Z12k_spmv_sdiabIdLb1ELb1ELi2ELi7ELi3EE")
    assert "wave_shr:1" in b and "wave_shl:1" in b
    assert count(b, "buffer_load_dwordx2") >= 10            # 2 slices x (centre + 4 gathered slots) on the class path
