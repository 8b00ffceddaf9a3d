"""CPU-only: host-side miner planning, RNG-stream equivalence, and the data-parallel bucket over gloo (world 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _det import det_int
from oracle import metric as OM


def test_triplet_plan_matches_oracle_counts():
    from hpcs_amd.miner.loss_and_miner_utils import TripletPlan
    lab = torch.cat([torch.zeros(3000), torch.ones(800), torch.full((296,), 2), torch.full((1,), 4)]).long()
    plan = TripletPlan.from_labels(lab, 50, 1.2)
    counts = torch.bincount(lab)
    want = OM.triplets_per_anchor(counts[counts > 0], 50, 1.2)
    got = [int(k) for k, c in zip(plan.k, plan.counts) if c > 0]
    assert got[:3] == want[:3] and got[3] == 0              # singleton class samples nothing
    assert plan.T == sum(int(c) * int(k) for c, k in zip(plan.counts, plan.k))
    torch.manual_seed(3)
    a, p, n = OM.balanced_triplets(lab, 50, 1.2)
    assert a.numel() == plan.T
    # the host draw stream is the reference's stream
    torch.manual_seed(3)
    hp, hn = plan.host_draws()
    assert hp.numel() == hn.numel() == plan.T
    torch.manual_seed(3)
    assert torch.equal(torch.randint(0, 2999, (hp[:150000].numel(),)), hp[:150000])


def test_plan_edge_cases():
    from hpcs_amd.miner.loss_and_miner_utils import TripletPlan
    assert TripletPlan.from_labels(torch.zeros(16).long(), 5, 1.2).T == 0           # single class: no negatives
    assert TripletPlan.from_labels(torch.arange(8), 5, 1.2).T == 0                  # all singletons
    p = TripletPlan.from_labels(torch.tensor([3, 3, 7, 7, 7]), 2, 0.0)
    assert p.C == 8 and p.T == 2 * 2 + 3 * 2 and int(p.toff[-1]) == p.T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hpcs_amd.dp import FlatGradBucket, broadcast_parameters, init_distributed
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    broadcast_parameters(net)                          # ... and are made identical
    bucket = FlatGradBucket(net.parameters())
    bucket.zero()
    x = det_int((8, 6), 7 + rank, 9).float()           # each rank owns a different shard of the batch
    net(x).square().sum().backward()                   # autograd accumulates straight into the flat bucket
    local = bucket.flat.clone()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    bucket.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = torch.allclose(bucket.flat, want, rtol=1e-6, atol=1e-7)
    params = torch.cat([p.detach().flatten() for p in net.parameters()])
    plist = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(plist, params)
    same = all(torch.equal(plist[0], q) for q in plist)
    out[rank] = bool(ok and same)
    dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_bucket_guards_against_detached_grads():
    from hpcs_amd.dp import FlatGradBucket
    lin = torch.nn.Linear(3, 2)
    b = FlatGradBucket(lin.parameters())
    b.zero()
    lin.weight.grad = torch.zeros_like(lin.weight)      # what zero_grad(set_to_none=True)+backward would leave
    with pytest.raises(RuntimeError):
        b.zero()


def test_gc_settle_is_opt_in_and_runs_once(monkeypatch):
    """hpcs_amd/step.py::_gc_settle: nothing unless HPCS_GC_FREEZE=1 (no hidden process-wide gc.freeze() from
    library code; bench.py opts in for its own process); with it, one collection + gc.freeze() per process."""
    import gc
    from hpcs_amd import step as S
    before = gc.get_freeze_count()
    for off in (None, "0"):
        monkeypatch.setattr(S, "_GC_SETTLED", False)
        if off is None:
            monkeypatch.delenv("HPCS_GC_FREEZE", raising=False)
        else:
            monkeypatch.setenv("HPCS_GC_FREEZE", off)
        S._gc_settle()
        assert gc.get_freeze_count() == before and S._GC_SETTLED
    monkeypatch.setattr(S, "_GC_SETTLED", False)
    monkeypatch.setenv("HPCS_GC_FREEZE", "1")
    try:
        S._gc_settle()
        frozen = gc.get_freeze_count()
        assert frozen > before
        S._gc_settle()                                   # second call: no-op
        assert gc.get_freeze_count() == frozen
    finally:
        gc.unfreeze()


def test_bench_launcher_starts_fresh_rank_processes_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher environment starts its own ranks (fresh children under
    torch.distributed.run, never an exec) and relays their status.  Here there is no GPU: every rank must come up, join the
    gloo group and then stop at bench.py's own "needs an MI355X" exit -- a non-zero status the parent passes on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("CPU-side launcher test (tests/test_gpu_dp.py runs the real thing on a GPU)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
