"""Kernels that prefetch with asm-issued global loads (conv3x3_f16x2.hip, conv1d_f16x2.hip): hipcc does not know those loads are
asynchronous, so a register copy it places between a load and its asm wait reads stale data, and a load still in flight when
its register is reused clobbers the new value (both happened while these kernels were written: a v_mov of the loop-carried halo
registers in front of the s_waitcnt; the weight ring's last refills landing in the epilogue's registers).  tools/asm_load_audit.py
screens the compiled ISA for both; this test compiles the two files for gfx950 (no GPU needed) and requires a clean report."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "text-to-sound-synthesis_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,allowed", [("conv3x3_f16x2.hip", 0), ("conv1d_f16x2.hip", 6)])
def test_no_use_of_registers_with_asm_loads_in_flight(tmp_path, src, allowed):
    out = str(tmp_path / "k.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S",
                        "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), out], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    n = int(a.stdout.strip().splitlines()[-1].split(":")[1])
    reads = [l for l in a.stdout.splitlines() if l.startswith("HAZARD line")]
    # conv1d: the linear scan meets the loop's halo-request block (laid out in front of the loop body) before the waits that
    # precede it at run time and reports its address temporaries as overwrites of ring registers: 6 known false positives,
    # none of them a READ of a pending register
    assert not reads and n <= allowed, a.stdout


def test_the_audit_sees_a_copy_in_front_of_the_wait(tmp_path):
    f = tmp_path / "t.s"
    f.write_text("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[66:69], v2, s[18:19]\n\t;;#ASMEND\n\tv_mov_b64_e32 v[124:125], v[68:69]\n"
                 "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mul_f32_e32 v126, 0x3e4ccccd, v66\n\tv_mov_b32_e32 v70, v1\n"
                 "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[80:83], v2, s[18:19]\n\t;;#ASMEND\n\tv_mov_b32_e32 v81, v1\n")
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_load_audit.py"), str(f)], capture_output=True, text=True)
    assert "HAZARD line 4" in a.stdout and "HAZARD (overwrite) line 13" in a.stdout and a.stdout.strip().endswith("hazards: 2")
