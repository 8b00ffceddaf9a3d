"""Standalone timings of the VN point-wise kernels at the step's shapes (no side-stream contention): us, MB, TB/s per kernel."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from hpcs_amd._lib import call

dev = "cuda"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def layer(B, Cout, Cd, L, tag, flags=3):
    Ctot = Cout + Cd
    PD = torch.randn(B * Ctot * 3 * L, device=dev)
    g = torch.randn(B * Cout * 3 * L, device=dev)
    mean = torch.rand(Cout, device=dev) + 1.0
    rstd = torch.rand(Cout, device=dev) + 0.5
    gamma = torch.rand(Cout, device=dev) + 0.5
    beta = torch.randn(Cout, device=dev) * 0.1
    red = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    sums = torch.zeros(2 * Cout + Cout, dtype=torch.float64, device=dev)
    GPD = torch.empty_like(PD)
    out = torch.empty_like(g)
    nb_pd, nb_g = PD.numel() * 4, g.numel() * 4
    def stats():
        sums.zero_(); call("hpcs_vn_norm_stats", PD, B, Cout, Ctot, L, sums, 1)
    def fwd():
        call("hpcs_vn_apply_fwd", PD, mean, rstd, gamma, beta, B, Cout, Cd, L, 0, 0.2, flags, out)
    def reduce():
        red.zero_(); call("hpcs_vn_bwd_reduce", PD, g, mean, rstd, gamma, beta, B, Cout, Cd, L, 0, 0.2, flags, red, 1)
    def apply():
        call("hpcs_vn_bwd_apply", PD, g, mean, rstd, gamma, beta, red, 1, B, Cout, Cd, L, 0, 0.2, flags, GPD, None, None)
    for nm, fn, nbytes in (("norm_stats", stats, nb_pd * Cout // Ctot), ("apply_fwd", fwd, nb_pd + nb_g), ("bwd_reduce", reduce, nb_pd + nb_g),
                           ("bwd_apply", apply, 2 * nb_pd + nb_g)):
        us = timeit(fn)
        print(f"{tag:28s} {nm:11s} {us:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)

layer(32, 341, 1, 1024, "conv6 341+1 L=1024")
layer(32, 341, 341, 1024, "std.vn1 341+341 L=1024")
layer(32, 170, 170, 1024, "std.vn2 170+170 L=1024")
layer(32, 21, 21, 20480, "edge 21+21 L=20480")
