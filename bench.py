#!/usr/bin/env python3
"""Headline benchmark: HPCS training-step throughput (point-clouds/s) on MI355X + kNN / GEMM rooflines.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of the hot path on one synthetic batch already resident in HBM:
SO(3) rotation -> VN_DGCNN_partseg (3 kNN graphs + EdgeConv + VN tail + head, train mode, dropout 0.5) -> ExpMap
-> class-balanced triplet mining -> HypHC + CosFace losses -> backward -> (gradient all-reduce) -> RAdam.
Workload = BASELINE.json configs[1]: ShapeNet-like, fixed_points=1024, k=20, E=32/32, batch=32 PER GPU (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this process is the benchmark and nothing else: opt in to the step driver's one-off gc.collect() + gc.freeze() (CPython's
# first full collection of the start-up heap otherwise lands as a 40-100 ms host stall inside a 20-step window)
os.environ.setdefault("HPCS_GC_FREEZE", "1")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16), 16x the fp32-input MFMA
GEMM_X3 = os.environ.get("HPCS_GEMM_X3", "1") != "0"      # large dense GEMMs as six bf16-piece products (csrc/gemm_x3.hip)
VALU_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md:41 fp32 vector peak (packed v_pk_fma_f32; 78.6 with scalar v_fma_f32 only)
L2_GATHER_PEAK_GBS = 17800.0  # MI355X_MICROARCH.md "Indexed rows": L2-resident row gather, 16.8-18.8 TB/s chip-wide


def synthetic_batch(B, N, num_class, num_categories, seed, labels="voronoi"):
    """N(0,I) clouds normalised to the unit ball, uniform categories, part labels either
    "voronoi"   : Voronoi cells of num_class seeded centroids (spatially coherent parts; the benchmark's labels), or
    "invariant" : rotation-INVARIANT parts -- radial shells x local-density halves (equal-mass bins over the batch).  A
                  rotation-invariant extractor fed SO(3)-augmented clouds cannot learn absolute-position (Voronoi) parts
                  (accuracy stays at chance); these it can, so training curves / trajectory tests use them."""
    pts = torch.empty(B, N, 3)
    for b in range(B):
        g = torch.Generator().manual_seed(1234 + seed * 100003 + b)
        p = torch.randn(N, 3, generator=g)
        p = p - p.mean(0, keepdim=True)
        pts[b] = p / p.norm(dim=1).max()
    g = torch.Generator().manual_seed(99 + seed)
    if labels == "invariant":
        n_shell = max(num_class // 2, 1)
        r = pts.norm(dim=2).reshape(-1)                                         # [B*N]
        d = torch.cdist(pts, pts)                                               # [B,N,N]
        dens = d.topk(9, dim=2, largest=False).values[:, :, -1].reshape(-1)     # distance to the 8th neighbour
        shell = (r.argsort().argsort().float() / r.numel() * n_shell).long().clamp_(max=n_shell - 1)
        tgt = torch.zeros_like(shell)
        for s_ in range(n_shell):                                               # density halves inside every shell
            m = shell == s_
            med = dens[m].median()
            tgt[m] = 2 * s_ + (dens[m] > med).long()
        tgt = tgt.clamp_(max=num_class - 1)
    elif labels == "voronoi":
        cent = torch.randn(num_class, 3, generator=g)
        cent = cent / cent.norm(dim=1, keepdim=True) * torch.rand(num_class, 1, generator=g) ** (1 / 3) * 0.8
        tgt = torch.cdist(pts.reshape(-1, 3), cent).argmin(dim=1)
    else:
        raise ValueError("labels must be 'voronoi' or 'invariant'")
    present = torch.bincount(tgt, minlength=num_class)
    missing = (present == 0).nonzero().flatten().tolist()
    for i, c in enumerate(missing):                      # every class non-empty (round-robin fallback)
        tgt[(i * 7919) % tgt.numel()] = c
    cat = torch.randint(0, num_categories, (B,), generator=g)
    return pts, cat, tgt.view(B, N)


def hierarchy_for(num_class):
    """Level lists [level][branch][class ids] for the hierarchical (PartNet) loss.  C = 39 / 51: the lists the reference's
    hierarchy parser produced for PartNet-format trees with Chair's / Table's level sizes (tests/golden/
    hierarchy_lists.json, made by tests/golden/make_golden_r2.py; the PartNet label files themselves are dataset files,
    absent here).  Other C: three nested levels of consecutive runs (groups of 8, groups of 3, leaves)."""
    path = os.path.join(ROOT, "tests", "golden", "hierarchy_lists.json")
    if os.path.exists(path):
        for v in json.load(open(path)).values():
            if v["num_class"] == num_class:
                return v["hierarchy_list"]
    C = num_class
    return [[list(range(i, min(i + 8, C))) for i in range(0, C, 8)],
            [list(range(i, min(i + 3, C))) for i in range(0, C, 3)], [[c] for c in range(C)]]


# BASELINE.json configs (per-GPU shapes); "2" is the configuration the metric is quoted on
PRESETS = {
    "1": dict(batch=4, points=256, k=10, num_class=4, num_categories=16, hierarchical=False,
              name="BASELINE configs[0] geometry: ShapeNet single category, fixed_points=256, k=10, batch=4"),
    "2": dict(batch=32, points=1024, k=20, num_class=50, num_categories=16, hierarchical=False,
              name="BASELINE configs[1]: ShapeNet-like fixed_points=1024, k=20, batch=32 per GPU"),
    "3": dict(batch=32, points=1024, k=20, num_class=39, num_categories=1, hierarchical=True,
              name="BASELINE configs[2]: PartNet Chair level-3 (39 parts, one category, hierarchical loss), "
                   "fixed_points=1024, k=20, batch=32 per GPU"),
    "5": dict(batch=16, points=2048, k=40, num_class=51, num_categories=1, hierarchical=True,
              name="BASELINE configs[4] per-GPU shape: PartNet Table level-3 (51 parts, hierarchical loss), "
                   "fixed_points=2048, k=40, batch=16 per GPU"),
}


def build_step(args, device):
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    from hpcs_amd.step import HypHCStep
    torch.manual_seed(0)
    feat = VN_DGCNN_partseg(in_channels=3, out_features=args.embedding, k=args.k, dropout=0.5, pooling="mean",
                            num_categories=args.num_categories)
    hier = None
    if args.hierarchical:
        hier = hierarchy_for(args.num_class)
    step = HypHCStep(feat, ExpMap(), hyp_size=args.embedding, num_class=args.num_class,
                     num_categories=args.num_categories, lr=0.05, margin=0.35, t_per_anchor=50, fraction=1.2,
                     temperature=0.05, trade_off=0.1, miner=True, cosface=True, train_rotation="so3", rng="device",
                     hierarchical=args.hierarchical, hierarchy_list=hier)
    step.to(device).train()
    return step.configure()


def pmc_traffic(kernel_substr):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (scripts/pmc_traffic.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    hit = [(v["hbm_bytes_per_launch"], v["launches"]) for k_, v in tab.items() if kernel_substr in k_]
    if not hit:
        return None
    return round(sum(b * n for b, n in hit) / sum(n for _, n in hit))


EXTRA_STEPS = 3
HOST_BURST = 3               # steps enqueued back to back from an idle queue to time the host side of a step
EVENT_STEPS = 2              # dominant-kernel event pairs in this many of the timed steps (evenly spaced)
EXTRA_NOTE = f"HIP events in {EXTRA_STEPS} further steps after the timed region"


def is_small_gemm(tag):
    """hpcs_gemm_tn's dispatch (gemm.hip): fewer than 32 tiles of 128x128 and a short reduction -> per-element kernel."""
    M, Nc, K, no, ni = tag[7], tag[8], tag[9], tag[10], tag[11]
    tiles = -(-M // 128) * -(-Nc // 128) * no * ni
    return tiles < 32 and Nc >= 16 and K <= 4096


def is_big_mfma_gemm(tag):
    """launches of gemm_tn_kernel<128,128,16,2,2>"""
    return tag[7] > 64 and not is_small_gemm(tag)


def gemm_bytes(tag):
    M, Nc, K, no, ni = tag[7], tag[8], tag[9], tag[10], tag[11]
    return 4.0 * (M * K + (K + M) * Nc * no * ni)


def gemm_flops(tag):
    # hpcs_gemm_tn ints: (lda, ldb, sBo, sBi, ldc, sCo, sCi, M, Nc, K, n_outer, n_inner, accumulate)
    M, Nc, K, no, ni = tag[7], tag[8], tag[9], tag[10], tag[11]
    return 2.0 * M * Nc * K * no * ni


def cpu_baseline(args, Bc=None, points=None, k=None, num_class=None, timed_steps=None):
    """The oracle's full step (fwd + loss + bwd + RAdam) on the host cores, on a bounded sample: SURVEY 8(d) asks for the
    metric's geometry at a reduced batch (B=8) -- one timed step of it is ~20 s on a 16-core share -- with the thread pools
    and torch paged in by a warm-up step of a SMALL geometry first."""
    from oracle import extractor as OE, hyperbolic as OH, metric as OM
    points = points or args.points
    k = k or args.k
    num_class = num_class or args.num_class
    if Bc is None:
        Bc = 8 if points >= 1024 else args.batch
    if timed_steps is None:
        timed_steps = 3              # (r03: one step -- a single sample with a +-12 % spread run to run)
    # the GPU box gives one GPU a 16-core share although os.cpu_count() reports the whole host
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle step B={Bc} N={points} k={k} on {cores} threads ...", file=sys.stderr, flush=True)
    torch.manual_seed(0)
    shapes = OE.param_shapes(args.embedding, 16)
    P = {}
    for k_, shp in shapes.items():
        if k_.endswith("num_batches_tracked"):
            P[k_] = torch.zeros((), dtype=torch.int64)
        elif k_.endswith("running_var") or (k_.endswith("weight") and len(shp) == 1):
            P[k_] = torch.ones(shp)
        elif k_.endswith("running_mean") or k_.endswith("bias"):
            P[k_] = torch.zeros(shp)
        else:
            P[k_] = torch.randn(shp) / (shp[1] ** 0.5)
    learn = [k_ for k_, v in P.items() if v.is_floating_point() and "running" not in k_ and not k_.startswith("bn")]
    for k_ in learn:
        P[k_].requires_grad_(True)
    OE.resolve_aliases(P)
    W = torch.randn(args.embedding, num_class, requires_grad=True)
    scale = torch.tensor([1e-3], requires_grad=True)
    params = [P[k_] for k_ in learn] + [W, scale]
    states = [{} for _ in params]
    group = {"betas": (0.9, 0.999), "eps": 1e-8, "lr": 0.05, "weight_decay": 0, "amsgrad": False, "step": 0}

    def one(B_, N_, k_nn, seed):
        pts, cat, tgt = synthetic_batch(B_, N_, num_class, 16, seed=seed)
        lab = torch.eye(16)[cat]
        for p_ in params:
            p_.grad = None
        x = pts.transpose(2, 1).contiguous()
        masks = [torch.empty(B_, 256, N_).bernoulli_(0.5) * 2 for _ in range(2)]
        xe = OE.vn_dgcnn_partseg(x, lab, P, k=k_nn, training=True, drop_masks=masks)
        xp = OH.expmap0(xe).reshape(-1, args.embedding)
        lh, lm, _ = OM.metric_hyperbolic_loss(xp, tgt.reshape(-1), scale, W, margin=0.35, t_per_anchor=50, fraction=1.2,
                                              temperature=0.05, cosface=True)
        (lm + 0.1 * lh).backward()
        OM.radam_step(params, [p_.grad for p_ in params], states, group)

    t0 = time.perf_counter()
    one(2, min(points, 256), min(k, 10), 6)         # warm-up on a small geometry (pages torch in, starts the thread pools)
    print(f"[bench] cpu_baseline warm-up step {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    per = []
    for i in range(timed_steps):
        t1 = time.perf_counter()
        one(Bc, points, k, 7)
        per.append(time.perf_counter() - t1)
        print(f"[bench] cpu_baseline step {i + 1}/{timed_steps} at {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    dt = sorted(per)[len(per) // 2]                                   # median step
    return {"value": round(Bc / dt, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "sample": f"oracle full step (fwd+loss+bwd+RAdam), B={Bc}, N={points}, k={k}, E={args.embedding}, "
                      f"C={num_class}, median of {timed_steps} timed step(s) after a small-geometry warm-up, torch "
                      f"{torch.get_num_threads()} threads",
            "s_per_step": round(dt, 3), "s_per_step_all": [round(v, 3) for v in per],
            "value_min_max": [round(Bc / max(per), 4), round(Bc / min(per), 4)]}


def knn_steady_state(dev, N, k, B=4096, D=3, reps=5):
    """at 32 clouds the kNN launch is in the launch-latency regime, so the kernel is also timed at B=4096
    (D=3: 50 MB of points in, 0.67 GB of int64 indices out) with HIP events on the launch stream."""
    from hpcs_amd import _lib
    x = torch.randn(B, D, N, device=dev)
    idx = torch.empty(B, N, k, dtype=torch.int64, device=dev)
    nbytes = _lib.call("hpcs_knn_workspace", B, N)
    ws = torch.empty(nbytes // 4, device=dev)
    _lib.call("hpcs_knn", x, B, D, N, k, idx, ws, nbytes)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.call("hpcs_knn", x, B, D, N, k, idx, ws, nbytes)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    byts, flops = B * N * (4 * D + 8 * k), 2.0 * D * N * N * B
    return {"B": B, "D": D, "N": N, "k": k, "ms_per_launch": round(t * 1e3, 3), "achieved_GBs": round(byts / t / 1e9, 1),
            "frac_hbm": round(byts / t / 1e9 / HBM_PEAK_GBS, 4), "fp32_tflops": round(flops / t / 1e12, 2),
            "frac_valu": round(flops / t / 1e12 / VALU_F32_PEAK_TF, 4),
            "clouds_per_s": round(B / t)}


def launch_ranks(n):
    """One process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on this script with the same
    arguments, rendezvous on 127.0.0.1 at a free port.  The children are fresh interpreters (subprocess); the parent only
    waits, so nothing that has initialised the GPU is ever replaced by another program.  Returns the launcher's exit code;
    rank 0's JSON line reaches stdout through the inherited descriptor."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(len(os.sched_getaffinity(0)) // n, 16))))
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(PRESETS), default=None,
                    help="BASELINE.json configuration preset (default: 2, the one the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=None, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--embedding", type=int, default=32)
    ap.add_argument("--num-class", type=int, default=None)
    ap.add_argument("--num-categories", type=int, default=None)
    ap.add_argument("--hierarchical", action="store_true", default=None,
                    help="PartNet-style hierarchical CosFace (configs 3 / 5)")
    ap.add_argument("--settle-min", type=int, default=30,
                    help="the settle rule may not fire before this many settle steps (the kept count is flat at first)")
    ap.add_argument("--settle-max", type=int, default=80,
                    help="upper bound of the untimed steps run until the miner's kept-triplet count is steady")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()
    preset = PRESETS[args.config or "2"]
    custom = []
    for key in ("batch", "points", "k", "num_class", "num_categories", "hierarchical"):
        if getattr(args, key) is None:
            setattr(args, key, preset[key])
        elif getattr(args, key) != preset[key]:
            custom.append(f"{key}={getattr(args, key)}")
    workload = preset["name"] if not custom else \
        "custom shape (" + ", ".join(custom) + f") on top of config {args.config or '2'}: fixed_points={args.points}, " \
        f"k={args.k}, batch={args.batch} per GPU, parts={args.num_class}" + (", hierarchical loss" if args.hierarchical else "")
    workload += f"; eucl/hyp_embedding={args.embedding}, vn_dgcnn_partseg + ExpMap + miner + HypHC + " \
                f"{'hierarchical ' if args.hierarchical else ''}CosFace + RAdam"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process (which has not touched the GPU) starts the N ranks
        # itself -- fresh child processes under torch.distributed.run, never an exec -- relays their output and exits
        # with their status
        raise SystemExit(launch_ranks(args.gpus))

    from hpcs_amd import _lib
    from hpcs_amd.dp import broadcast_parameters, init_distributed
    import torch.distributed as dist

    rank, local, world = init_distributed()
    # os.cpu_count() reports the whole host (256) while one GPU's share is 16 cores: cap torch's host threads, and share
    # the visible cores between the ranks of THIS node (the step's host-side torch ops are tiny)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)) // max(local_world, 1), 16)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (hpcs_amd has no CPU fallback)")
    if os.environ.get("HPCS_SINGLE_DEVICE_REHEARSAL"):      # several ranks sharing cuda:0 (gloo), code-path rehearsal only
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    step = build_step(args, device)
    broadcast_parameters(step)
    pts, cat, tgt = synthetic_batch(args.batch, args.points, args.num_class, args.num_categories, seed=rank)
    pts_d, cat_d, tgt_d = pts.to(device), cat.to(device), tgt.to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        return step.training_step(pts_d, cat_d, tgt_d, targets_host=tgt)

    # Order: warm-up, settle, THEN the step's auto mode (capture forms against eager), so that the mode is chosen on the
    # workload the timed region runs (r03: chosen on the pre-ramp workload, a replay that had measured 1.5 % ahead of eager
    # ran the settled steps 3-4 % behind it).  The step's own eager warm-up count is held off until the settle is done.
    hold_warmup, step.graph_warmup = step.graph_warmup, 1 << 30
    for _ in range(args.warmup):
        one_step()
    # Steady state: training lets more triplets through the miner's filter between about step 15 and step 40 (8.2 M -> 13 M
    # kept of 16.9 M at config 2; flat before that), and the loss kernel's time follows; the timed region starts once the
    # kept count has moved by less than 1 % over 5 steps, no earlier than SETTLE_MIN steps in (the flat start satisfies the
    # 1 % rule too) and no later than SETTLE_MAX.  Untimed, reported; every rank runs the same number of steps.
    n_settle, prev_kept, settle_hist = 0, None, []
    while n_settle < args.settle_max:
        for _ in range(5):
            one_step()
        n_settle += 5
        kept_now = float(step.metric_hyp_loss.last_kept.item())
        settle_hist.append(int(kept_now))
        done = (prev_kept is not None and abs(kept_now - prev_kept) <= 0.01 * max(kept_now, 1.0)
                and n_settle >= min(args.settle_min, args.settle_max))
        prev_kept = kept_now
        if world > 1:
            flag = torch.tensor([1.0 if done else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item() > 0.5)
        if done:
            break
    step.graph_warmup = hold_warmup                      # the eager steps run so far count towards it
    n_tune = 0
    while step.graph_mode() is None and n_tune < 24:      # auto mode: capture + timing of replay against eager (untimed)
        one_step()
        n_tune += 1
    mode = step.graph_mode() or "eager"
    mode_label = f"graph captured {step.graph_variant()}" if mode == "graph" else mode
    barrier()
    timed = not args.no_kernel_timing
    # HIP events (on the launch stream) around every launch of the two heaviest kernels, live in the timed region:
    # hpcs_gemm_tn calls that dispatch gemm_tn_kernel<128,128,16,2,2> (M > 64 and >= 32 tiles, see gemm.hip) and
    # hpcs_hyphc_sampled.  An event pair is a barrier packet on each side; r03 measured what that does to an EAGER two-stream
    # step: with pairs in every 4th step the same box gave 10.9 ... 16.2 ms per step run to run (10.6-11.0 without any), so
    # the pairs are recorded in EVENT_STEPS of the timed steps only, from pre-created event objects -- still inside the
    # timed region, still every launch of those kernels in those steps.  When the steady state replays a hipGraph those
    # sampled steps run eagerly (events cannot be wrapped around the nodes of a replayed graph).
    sampled = [False]
    sample_steps = sorted({(j + 1) * args.steps // (EVENT_STEPS + 1) for j in range(EVENT_STEPS)}) if timed else []
    if timed:
        # (only the step's longest kernel gets its pair inside the timed region -- ONE pair per sampled step; the 14 launches of
        # the big GEMM class are timed with the other secondary kernels in EXTRA_STEPS further steps after it: their 28
        # barrier packets per step are what made the timed region slow and erratic)
        _lib.PROFILE = {"hpcs_hyphc_sampled": []}
        _lib.PROFILE_FILTER = lambda name, tg: sampled[0]
        _lib.EVENT_POOL = [torch.cuda.Event(enable_timing=True) for _ in range(64 * max(len(sample_steps), 1))]
    kept_hist = []
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        sampled[0] = i_step in sample_steps
        if sampled[0] and mode == "graph":
            losses = step._eager_step(pts_d, cat_d, tgt_d, tgt)
        else:
            losses = one_step()
        if i_step in (0, args.steps - 1):
            kept_hist.append(step.metric_hyp_loss.last_kept.clone())       # device scalar: no sync
    barrier()
    dt = time.perf_counter() - t0
    sampled[0] = False
    # host cost of enqueueing one step, measured on a short burst from an idle queue (over the whole timed loop the host
    # is throttled by the ring of per-step parameter slots once it runs several steps ahead of the GPU)
    t1 = time.perf_counter()
    for _ in range(HOST_BURST):
        one_step()
    host_ms = (time.perf_counter() - t1) / HOST_BURST * 1e3
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    _lib.PROFILE_FILTER = None
    _lib.EVENT_POOL = None
    prof_extra = {}
    if timed:                  # every rank runs them: the step contains the gradient all-reduce
        # the secondary kernels (EdgeConv GEMM, kNN) are timed in EXTRA_STEPS further EAGER steps OUTSIDE the timed
        # region, so that their event pairs do not perturb the reported throughput
        _lib.PROFILE = {"hpcs_knn": [], "hpcs_gemm_tn": []}
        for _ in range(EXTRA_STEPS):
            step._eager_step(pts_d, cat_d, tgt_d, tgt)
        torch.cuda.synchronize()
        prof_extra, _lib.PROFILE = _lib.PROFILE, None
    # multi-rank diagnostics: per-rank step time, host enqueue cost and the all-reduce's event span
    dp_diag = None
    if world > 1:
        ar = []
        real_ar = step.bucket.all_reduce_mean

        def timed_ar(*a_, **k_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            real_ar(*a_, **k_)
            e1.record()
            ar.append((e0, e1))
        step.bucket.all_reduce_mean = timed_ar
        for _ in range(EXTRA_STEPS):
            step._eager_step(pts_d, cat_d, tgt_d, tgt)
        torch.cuda.synchronize()
        step.bucket.all_reduce_mean = real_ar
        mine = torch.tensor([dt / args.steps * 1e3, host_ms, sum(a_.elapsed_time(b_) for a_, b_ in ar) / max(len(ar), 1)],
                            device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        cols = torch.stack(allr).cpu()
        dp_diag = {"per_rank_ms_per_step": [round(float(v), 3) for v in cols[:, 0]],
                   "per_rank_host_enqueue_ms": [round(float(v), 3) for v in cols[:, 1]],
                   "per_rank_allreduce_span_ms": [round(float(v), 3) for v in cols[:, 2]],
                   "allreduce_bytes": int(step.bucket.numel * 4),
                   "note": "all-reduce span = HIP events around dist.all_reduce + the 1/world scale on the step's stream (it "
                           "includes waiting for the slowest rank); measured in extra eager steps after the timed region"}
        dist.barrier()
    kept_window = [int(k_.item()) for k_ in kept_hist]

    t = torch.tensor([dt], device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(losses["total_loss"])

    if rank != 0:
        return
    clouds = args.batch * world * args.steps
    cps = clouds / dt
    T_s = int(step.metric_hyp_loss.last_T)
    out = {
        "metric": "point-clouds/sec training step (N=%d,k=%d)" % (args.points, args.k),
        "value": round(cps, 2),
        "unit": "point-clouds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": args.batch * world, "num_class": args.num_class,
                   "parallelism": f"dp{world} (per-replica BN and mining, one flat-gradient all-reduce)",
                   "step_mode": f"{mode_label} (HPCS_GRAPH={os.environ.get('HPCS_GRAPH', 'auto')}"
                                + (f"; measured ms/step per candidate: {step._tune['measured_ms']}"
                                   if getattr(step, '_tune', None) and step._tune.get('measured_ms') else "") + ")",
                   "untimed_steps_before_the_timed_region": args.warmup + n_tune + n_settle,
                   "steady_state": f"{n_settle} untimed settle steps after warm-up, before mode tuning: kept triplets every 5 steps "
                                   f"{settle_hist} (timed region starts once they move < 1 % per 5 steps)",
                   "triplet_rng": "device counter RNG (statistically equivalent to the reference's CPU torch.randint)",
                   "dense_gemm": ("fp32 operands and results; the large dense layers (M > 64) form each fp32 product from three "
                                  "exact bf16 pieces per operand -- six piece products on the bf16 matrix cores, fp32 accumulate "
                                  "(csrc/gemm_x3.hip): error against the fp64 product no larger than the fp32 fma-chain kernel's "
                                  "(tests/test_gpu_round4.py); HPCS_GEMM_X3=0 restores the fma chain. kNN, EdgeConv mixes and "
                                  "every element-wise kernel are plain fp32") if GEMM_X3 else
                                 "fp32 fma-chain MFMA (HPCS_GEMM_X3=0)",
                   "metrics_in_step": "accuracy + IoU of the CosFace logits computed on device every step (base_hyp_hc.py:126-130), "
                                      "no host sync; last step: acc=%.4f iou=%.4f" % (float(losses["acc"]), float(losses["iou"])),
                   "final_total_loss": round(loss_val, 6),
                   "host_enqueue_ms_per_step": round(host_ms, 3),
                   "data_parallel": dp_diag,
                   "triplets_sampled_per_step_rank0": T_s,
                   "triplets_kept_first_and_last_timed_step": kept_window},
    }
    # step-level roofline (SURVEY 8d / BASELINE.md 4): clouds/s x algorithmic fwd+bwd FLOPs per cloud / fp32 MFMA peak
    flop_cloud = {(1024, 20): 18.2e9, (256, 10): 4.15e9, (2048, 40): 41.98e9}.get((args.points, args.k))
    if flop_cloud:
        out["roofline_step"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TF * world,
                                "achieved": round(cps * flop_cloud / 1e12, 2),
                                "frac": round(cps * flop_cloud / 1e12 / (MFMA_F32_PEAK_TF * world), 4),
                                "flop_per_cloud_fwd_bwd": flop_cloud,
                                "note": "SURVEY 8(d)'s un-hoisted count (the reference's op sequence); the build skips "
                                        "work by hoisting conv8's 2110 and vn1's 341 N-constant channels"}
    if timed and prof:
        def dur(lst):
            return [(e0.elapsed_time(e1) * 1e-3, tag) for e0, e1, tag in lst]
        n_sampled = max(len(sample_steps), 1)
        step_s = dt / args.steps
        hb = dur(prof["hpcs_hyphc_sampled"])
        kn = dur(prof_extra["hpcs_knn"])
        gm_extra = dur(prof_extra["hpcs_gemm_tn"])
        big = [(d, t_) for d, t_ in gm_extra if is_big_mfma_gemm(t_)]
        small = [(d, t_) for d, t_ in gm_extra if t_[7] <= 64 and not is_small_gemm(t_)]
        gm_flops = sum(gemm_flops(t_) for _, t_ in big)
        gm_time = sum(d for d, _ in big)
        hb_time = sum(d for d, _ in hb)
        rank_gemm = 1 if gm_time / EXTRA_STEPS >= hb_time / n_sampled else 2
        if GEMM_X3:
            # six bf16 MFMAs per fp32 product: the pipe executes 6x the algorithmic flops, so the ceiling of this form is the
            # bf16 peak / 6; `achieved` stays in algorithmic (fp32-equivalent) TFLOP/s so that it compares with earlier rounds
            eff_peak = MFMA_BF16_PEAK_TF / 6.0
            gemm_obj = {"kernel": "gemm_tn_x3_kernel (hpcs_gemm_tn, M > 64: VN dense layers + heads, forward and data-grad; "
                                  "fp32 products as six exact bf16-piece products on v_mfma_f32_32x32x16_bf16, fp32 accumulate)",
                        "bound": "mfma", "achieved": round(gm_flops / gm_time / 1e12, 3), "peak": round(eff_peak, 1),
                        "unit": "TFLOP/s (algorithmic fp32 flops; peak = bf16 dense peak / 6 piece products)",
                        "frac": round(gm_flops / gm_time / 1e12 / eff_peak, 4),
                        "executed_bf16_TFLOPs": round(6 * gm_flops / gm_time / 1e12, 1), "bf16_peak_TFLOPs": MFMA_BF16_PEAK_TF,
                        "vs_fp32_mfma_peak": round(gm_flops / gm_time / 1e12 / MFMA_F32_PEAK_TF, 4),
                        "traffic": pmc_traffic("gemm_tn_x3_kernel"),
                        "algorithmic_bytes_per_launch": round(sum(gemm_bytes(t_) for _, t_ in big) / len(big)),
                        "launches": len(big), "avg_launch_us": round(1e6 * gm_time / len(big), 2),
                        "share_of_step": round(gm_time / (step_s * EXTRA_STEPS), 4),
                        "rank_by_time": rank_gemm,
                        "note": "`peak` prices the matrix pipe at 2.4 GHz; on random operands the chip holds 1.5-1.95 GHz under "
                                "dense bf16 MFMA load (MI355X_MICROARCH.md 'DVFS give-back': its best plain bf16 GEMMs reach "
                                "1250-1480 TFLOP/s = 208-247 in these units), and the average is over all 14 launches of a step "
                                "including the launch-sized head layers; the four VN-sized launches run at 159-175 "
                                "(profiles/r04_gemm_x3_bench_with_dma_variant.txt)",
                        "measured": EXTRA_NOTE}
        else:
            gemm_obj = {"kernel": "gemm_tn_kernel<128,128,16,2,2> (hpcs_gemm_tn, M > 64: VN dense layers + heads, forward and "
                                  "data-grad)",
                        "bound": "mfma", "achieved": round(gm_flops / gm_time / 1e12, 3), "peak": MFMA_F32_PEAK_TF,
                        "unit": "TFLOP/s", "frac": round(gm_flops / gm_time / 1e12 / MFMA_F32_PEAK_TF, 4),
                        "traffic": pmc_traffic("gemm_tn_kernel<128"),
                        "algorithmic_bytes_per_launch": round(sum(gemm_bytes(t_) for _, t_ in big) / len(big)),
                        "launches": len(big), "avg_launch_us": round(1e6 * gm_time / len(big), 2),
                        "share_of_step": round(gm_time / (step_s * EXTRA_STEPS), 4),
                        "rank_by_time": rank_gemm,
                        "measured": EXTRA_NOTE}
        hyphc_obj = None
        if hb:
            E_ = args.embedding
            kept = kept_window[-1] if kept_window else T_s // 2
            avg = hb_time / len(hb)
            # the loss is gather-bound, T * (3*E*4 + 3*8) algorithmic bytes per step (three rows and three
            # indices per sampled triplet).  That is the contract's `achieved`; what physically binds the kernel is not HBM
            # (the 4 MB embedding table is L2-resident: PMC sees ~1.9 GB per launch, nearly all WRITE = the float atomics of
            # the negative rows) but the chip-wide float-atomic rate, 1.3 TB/s of added bytes: kept * E * 4 bytes.
            alg = T_s * (3 * 4 * E_ + 3 * 8)
            atomic_floor_s = kept * 4.0 * E_ / 1.3e12
            l2_bytes = T_s * (2 * 4 * E_) + kept * (2 * 4 * E_)
            hyphc_obj = {"kernel": ("hyphc_sampled_kernel<8,true>" if E_ == 64 else f"hyphc_sampled4_kernel<{E_ // 8},true>") +
                                   " (hpcs_hyphc_sampled: triplets generated in-kernel, miner "
                                   "rule inline, loss + gradient in one sweep)",
                         "bound": "hbm", "achieved": round(alg / avg / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / avg / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("hyphc_sampled"),
                         "algorithmic_bytes_per_launch": alg, "launches": len(hb), "avg_launch_us": round(1e6 * avg, 2),
                         "share_of_step": round(hb_time / (step_s * n_sampled), 4), "rank_by_time": 3 - rank_gemm,
                         "triplets_sampled": T_s, "triplets_kept": kept,
                         "binding_ceiling": {"what": "chip-wide float-atomic rate on the negative rows' gradients: "
                                                     "kept * E * 4 B / 1.3 TB/s (MI355X_MICROARCH 'Global float atomics')",
                                             "floor_us": round(1e6 * atomic_floor_s, 1),
                                             "frac_of_ceiling": round(atomic_floor_s / avg, 4)},
                         "l2_gather": {"gathered_bytes_per_launch": l2_bytes, "achieved_GBs": round(l2_bytes / avg / 1e9, 1),
                                       "peak_GBs": L2_GATHER_PEAK_GBS, "frac": round(l2_bytes / avg / 1e9 / L2_GATHER_PEAK_GBS, 4)},
                         "measured": "HIP events around the entry point (one zero-fill launch + the kernel) in "
                                     f"{len(sample_steps)} of the {args.steps} steps of the timed region",
                         "note": "`achieved` follows SURVEY 8(d)'s algorithmic bytes against the HBM line as the bench contract "
                                 "asks; the kernel's real ceiling is `binding_ceiling`"}
        # `roofline` = the kernel with the largest share of the step; the other one keeps its own key
        if hyphc_obj is not None and rank_gemm == 2:
            out["roofline"], out["roofline_gemm"] = hyphc_obj, gemm_obj
        else:
            out["roofline"] = gemm_obj
            if hyphc_obj is not None:
                out["roofline_hyphc"] = hyphc_obj
        if small:
            sb = sum(gemm_bytes(t_) for _, t_ in small)
            st_ = sum(d for d, _ in small)
            out["roofline_edge_gemm"] = {"kernel": "gemm_tn_kernel<64,128,16,1,4> (EdgeConv channel mix, M <= 64)",
                                         "bound": "hbm", "achieved": round(sb / st_ / 1e9, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(sb / st_ / 1e9 / HBM_PEAK_GBS, 4),
                                         "traffic": pmc_traffic("gemm_tn_kernel<64"),
                                         "algorithmic_bytes_per_launch": round(sb / len(small)), "launches": len(small),
                                         "avg_launch_us": round(1e6 * st_ / len(small), 2),
                                         "share_of_step": round(st_ / (step_s * EXTRA_STEPS), 4), "measured": EXTRA_NOTE}
        if kn:
            # kNN: algorithmic bytes per launch = B*N*(4*D + 8*k)   (SURVEY 8d); tag = (B, D, N, k).  D = 3 (coordinates):
            # scores and selection are VALU work; D = 63 (features): the Gram contraction runs on the fp32 MFMA pipe, the
            # selection on VALU -- reported per D against the respective peak
            def knn_obj(sel, peak, bound):
                if not sel:
                    return None
                byts = sum(t_[0] * t_[2] * (4 * t_[1] + 8 * t_[3]) for _, t_ in sel)
                flops = sum(2.0 * t_[1] * t_[2] * t_[2] * t_[0] for _, t_ in sel)
                tm = sum(d for d, _ in sel)
                return {"bound": bound, "achieved": round(flops / tm / 1e12, 3), "unit": "TFLOP/s (2*D*N^2 per cloud)",
                        "peak": peak, "frac": round(flops / tm / 1e12 / peak, 4), "hbm_GBs": round(byts / tm / 1e9, 2),
                        "hbm_frac": round(byts / tm / 1e9 / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes_per_launch": round(byts / len(sel)), "launches": len(sel),
                        "avg_launch_us": round(1e6 * tm / len(sel), 2)}
            d3 = knn_obj([e for e in kn if e[1][1] <= 3], VALU_F32_PEAK_TF, "valu (scores + top-k selection)")
            d63 = knn_obj([e for e in kn if e[1][1] > 3], MFMA_F32_PEAK_TF, "mfma (Gram contraction) + valu selection")
            out["roofline_knn"] = {"kernel": "knn_sumsq_kernel + knn_mfma_*_kernel (hpcs_knn)",
                                   "note": "arithmetic intensity 36-313 flop/B is far above the HBM ridge (SURVEY 8d): HBM GB/s "
                                           "is reported beside the compute fraction for the BASELINE line, it cannot bind",
                                   "D3": d3, "D63": d63, "traffic": pmc_traffic("knn_mfma"), "measured": EXTRA_NOTE}
            if world == 1:
                out["roofline_knn"]["steady_state"] = knn_steady_state(device, args.points, args.k)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args)
        if (args.points, args.k) != (256, 10):
            # SURVEY 8(d) also asks for configs[0] (the reference's own CPU-runnable case) in full
            out["cpu_baseline_cfg1"] = cpu_baseline(args, Bc=4, points=256, k=10, num_class=4, timed_steps=3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
