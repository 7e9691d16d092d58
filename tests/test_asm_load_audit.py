"""Kernels that prefetch with asm-issued global loads (conv3x3_f16x2.hip, conv1d_f16x2.hip): hipcc does not know those loads are
asynchronous, so a register copy it places between a load and its asm wait reads stale data, and a load still in flight when
its register is reused clobbers the new value (both happened while these kernels were written: a v_mov of the loop-carried halo
registers in front of the s_waitcnt; the weight ring's last refills landing in the epilogue's registers).  tools/asm_load_audit.py
screens the compiled ISA for these patterns inside basic blocks (and for two asm loads in flight into ONE register: results the
compiler considers dead -- it then reuses that register at once); this test compiles the two files for gfx950 (no GPU needed) and requires a clean report."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "text-to-sound-synthesis_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["conv3x3_f16x2.hip", "conv1d_f16x2.hip"])
def test_no_use_of_registers_with_asm_loads_in_flight(tmp_path, src):
    out = str(tmp_path / "k.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S",
                        "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), out], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    assert a.stdout.strip().endswith("hazards: 0"), a.stdout


def test_the_audit_sees_a_copy_in_front_of_the_wait(tmp_path):
    f = tmp_path / "t.s"
    f.write_text("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[66:69], v2, s[18:19]\n\t;;#ASMEND\n\tv_mov_b64_e32 v[124:125], v[68:69]\n"
                 "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mul_f32_e32 v126, 0x3e4ccccd, v66\n\tv_mov_b32_e32 v70, v1\n"
                 "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[80:83], v2, s[18:19]\n\t;;#ASMEND\n\tv_mov_b32_e32 v81, v1\n"
                 "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[90:93], v2, s[18:19]\n\t;;#ASMEND\n"
                 "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[90:93], v3, s[18:19]\n\t;;#ASMEND\n")
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), str(f)], capture_output=True, text=True)
    assert "HAZARD line 4" in a.stdout and "HAZARD (overwrite) line 13" in a.stdout and "HAZARD (dead load) line 18" in a.stdout
    assert a.stdout.strip().endswith("hazards: 3")
