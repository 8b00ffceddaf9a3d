"""Builds hpcs_amd/csrc/libhpcs_mi355x.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhpcs_mi355x.so")
ARCH = "gfx950"
# per-file extra flags: knn.hip pins an exact fp32 rounding sequence (and keeps its MFMA results in VGPRs: every score is
# consumed by VALU code, an AGPR result costs one v_accvgpr_read each), decode.hip scipy's fp64 cosine sequence
EXTRA = {"knn.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"], "decode.hip": ["-ffp-contract=off"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
           [os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "hpcs_mi355x.h")]
    objs, jobs = [], []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, "-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
                   "-c", s, "-o", o] + EXTRA.get(src, [])
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
