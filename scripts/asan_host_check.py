#!/usr/bin/env python3
"""Host-side AddressSanitizer build of the C-ABI launch shims (sanitizers run on the CPU build only; GPU
ASan is not available on this pool).

Compiles every csrc/*.hip with `hipcc -fsanitize=address -fno-gpu-sanitize` (host code instrumented, device code as
usual), links /tmp/.../libhpcs_mi355x_asan.so and drives the entry points that return BEFORE any HIP call -- argument
validation, workspace / ticket-count queries, plan arithmetic -- through ctypes in a child process with the ASan runtime
preloaded.  These are the code paths that pack descriptors, size workspaces and choose kernels on the host.  Exit code 0 =
no report.  (No GPU needed; kernels are not launched.)
"""
import ctypes
import glob
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hpcs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

DRIVER = r'''
import ctypes, sys
L = ctypes.CDLL(sys.argv[1])
P, I, L64, Z, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
def fn(name, res, *args):
    f = getattr(L, name); f.restype = res; f.argtypes = list(args); return f
assert fn("hpcs_version", I)() >= 100
knn = fn("hpcs_knn", I, P, I, I, I, I, P, P, Z, P)
assert knn(None, 1, 3, 8, 2, None, None, 0, None) == -4
assert knn(1, 1, 3, 8, 9, 1, None, 0, None) == -1
assert knn(1, 1, 3, 128, 65, 1, None, 0, None) == -2
ws = fn("hpcs_gemm_nt_workspace", Z, I, I, I, I, I)
tk = fn("hpcs_gemm_nt_tickets", Z, I, I, I, I, I)
for shp in ((42, 42, 61440, 32, 1), (42, 21, 61440, 32, 1), (341, 682, 3072, 32, 1), (682, 682, 3072, 32, 1), (256, 2110, 1, 32, 1),
            (3, 170, 96, 4, 1), (1, 1, 1, 1, 1), (64, 256, 1024, 32, 1)):
    assert ws(*shp) > 0 and tk(*shp) >= 2, shp
assert ws(0, 1, 1, 1, 1) == 0 and tk(0, 1, 1, 1, 1) == 0
for name, args in (("hpcs_knn_workspace", (I, I)), ("hpcs_edge_gather_bwd_workspace", (I, I, I)),
                   ("hpcs_edge_grad_reduce_workspace", (I, I, I)), ("hpcs_complete_linkage_workspace", (I, I)),
                   ("hpcs_class_index_workspace", (I, I)), ("hpcs_triplet_filter_workspace", (L64,))):
    f = fn(name, Z, *args)
    for v in (1, 4, 32, 257):
        f(*([v] * len(args)))
nt = fn("hpcs_gemm_nt_fused", I, P, I, L64, L64, P, I, L64, L64, I, I, I, I, I, P, I, I, P, Z, P, Z, P)
assert nt(None, 1, 0, 0, None, 1, 0, 0, 1, 1, 1, 1, 1, None, 1, 0, None, 0, None, 0, None) == -4
tn = fn("hpcs_gemm_tn", I, P, I, P, I, L64, L64, P, I, L64, L64, I, I, I, I, I, P, I, P)
assert tn(None, 1, None, 1, 0, 0, None, 1, 0, 0, 1, 1, 1, 1, 1, None, 0, None) < 0
cm = fn("hpcs_class_metrics", I, P, P, I, I, P, P, P)
assert cm(None, None, 4, 4, None, None, None) == -4
assert cm(1, 1, 4, 5000, 1, 1, None) == -1
hs = fn("hpcs_hyphc_sampled", I, P, P, P, P, P, P, I, I, I, L64, ctypes.c_uint64, P, P, F, F, I, P, P, P, P, P, P, P, P)
assert hs(None, None, None, None, None, None, 4, 8, 32, 10, 1, None, None, 0.05, 0.0, 1, None, None, None, None, None, None, None, None) == -4
assert hs(1, 1, 1, 1, 1, 1, 4, 8, 24, 10, 1, None, 1, 0.05, 0.0, 1, 1, 1, None, None, None, None, None, None) == -2
dbg = fn("hpcs_debug_watch", I, P, P, I, I, P, P)
assert dbg(None, None, 16, 1, None, None) == -4 and dbg(1, 1, 6, 1, 1, None) == -1
print("asan driver ok")
'''


def main():
    out = tempfile.mkdtemp(prefix="hpcs_asan_")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    extra = {"knn.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"], "decode.hip": ["-ffp-contract=off"]}

    def cc(src):
        o = os.path.join(out, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC, "-O1", "-g", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fsanitize=address", "-fno-gpu-sanitize",
               "-shared-libsan", "-c", src, "-o", o] + extra.get(os.path.basename(src), [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-2000:])
        return o
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, srcs))
    lib = os.path.join(out, "libhpcs_mi355x_asan.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize",
                        "-shared-libsan", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        raise RuntimeError("ASan runtime not found under /opt/rocm/lib/llvm")
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97")
    r = subprocess.run([sys.executable, "-c", DRIVER, lib], capture_output=True, text=True, env=env, timeout=600)
    sys.stdout.write(r.stdout)
    if r.returncode or "AddressSanitizer" in r.stderr or "asan driver ok" not in r.stdout:
        sys.stderr.write(r.stderr[-4000:])
        return 1
    print(f"host ASan build of {len(srcs)} sources: no report ({lib})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
