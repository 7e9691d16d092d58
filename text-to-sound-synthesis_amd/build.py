"""Build recipe for libdiffsound_hip.so (gfx950 only; hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "obj")
LIB = os.path.join(HERE, "libdiffsound_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_f16x2.hip", "gemm_f16x2_ps.hip", "conv_f16x2.hip", "conv3x3_f16x2.hip", "conv1d_f16x2.hip", "melgan_fused.hip", "norm.hip", "attention.hip", "attention_bwd.hip", "attention_f16x2.hip", "sampler.hip", "misc.hip", "train.hip", "pack.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
         "-I", CSRC, "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
# kernels that must not touch scratch memory: a register demotion in one of them is a silent 3x slowdown
# (it happened once: accumulators of the f16x2 GEMM went to scratch when its epilogue grew a second store family)
NO_SCRATCH = ("ds_gemm_f16x2", "ds_attn_f16x2", "ds_gemm_kernel", "ds_sample_tail", "ds_conv2d_f16x2", "ds_conv3x3_f16x2",
              "ds_attn_bwd", "ds_melgan_rb", "ds_melgan_convt2", "ds_conv1d_f16x2")


def _code_only(text):
    """C / C++ source with comments and all whitespace removed (string literals are kept as they are)."""
    import re
    out = re.sub(r'//[^\n]*|/\*.*?\*/|("(?:\\.|[^"\\])*")', lambda m: m.group(1) or "", text, flags=re.S)
    return re.sub(r"\s+", "", out)


# what the dominant kernel of the sampling path (ds_gemm_f16x2_ps_kernel, gemm_f16x2_ps.hip) is compiled from
DOMINANT_KERNEL_SOURCES = ("gemm_f16x2_ps.hip", "gemm_f16x2_ps_epilogue.inc", "common.h")


def source_fingerprint(files=DOMINANT_KERNEL_SOURCES):
    """16 hex digits over the CODE (comments and whitespace stripped) of the sources the dominant kernel is compiled from
    (its .hip file and common.h; the public header only contributes enum values and grows with every new entry point, so
    it is left out).  Measurements that describe that kernel
    (profiles/*_pmc_*.json -> bench.py's roofline.traffic) carry it and are reported only while it still matches the
    tree -- a comment edit or a change to another kernel's file does not make them stale, a change to this kernel does."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(files):
        with open(os.path.join(CSRC, f), "r") as fh:
            h.update(f.encode() + b"\0" + _code_only(fh.read()).encode())
    return h.hexdigest()[:16]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _check_scratch(src, remarks):
    """Parse -Rpass-analysis=kernel-resource-usage remarks: 'Function Name: X' ... 'ScratchSize [bytes/lane]: N'."""
    import re
    name = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and int(m.group(1)) > 0 and any(k in name for k in NO_SCRATCH):
            raise RuntimeError("%s: kernel %s uses %s bytes/lane of scratch (register demotion) -- restructure it"
                               % (src, name, m.group(1)))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_f16x2_ps_epilogue.inc"),
               os.path.join(ROOT, "include", "diffsound_hip.h")]
    hipcc = _hipcc()

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
            _check_scratch(src, r.stderr)
            if verbose:
                print("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l))
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
