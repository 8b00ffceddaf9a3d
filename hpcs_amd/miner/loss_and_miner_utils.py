"""Mirror of hpcs/miner/loss_and_miner_utils.py:7-75 (class-balanced random triplet sampler) on HIP kernels.

Two RNG modes:
  rng="host"   : positives / negatives are drawn with torch.randint on the CPU default generator in exactly the
                 reference's order (per ascending label: n_a*k positive draws, then n_a*k negative draws) and mapped
                 to indices on the device -> identical triplets to the reference under the same torch seed.
  rng="device" : a counter-based generator inside the kernel (no host draws, no PCIe upload); statistically
                 equivalent, used by bench.py (stated there).
"""
import torch

from .. import _lib


class TripletPlan:
    """Host-side per-class plan: k_l = int(t * (max_count/n_l)^fraction) in fp32 tensor math like the reference."""

    def __init__(self, counts_cpu, t_per_anchor, fraction):
        counts = counts_cpu.to(torch.int64)
        C = counts.numel()
        self.C = C
        self.counts = counts
        M = int(counts.sum())
        self.M = M
        mx = torch.max(counts) if C else torch.tensor(0)
        ks, toff, T = [], [0], 0
        self.draw_specs = []          # (n_a, n_neg, T_c) for classes that sample, ascending label
        for c in range(C):
            n_a = int(counts[c])
            k = 0
            if n_a >= 2 and M - n_a >= 1:
                k = n_a if t_per_anchor is None else int(t_per_anchor * torch.pow((mx / n_a), fraction))
                k = max(k, 0)
            ks.append(k)
            T += n_a * k
            toff.append(T)
            if k > 0:
                self.draw_specs.append((n_a, M - n_a, n_a * k))
        self.k = torch.tensor(ks, dtype=torch.int32)
        self.toff = torch.tensor(toff, dtype=torch.int64)
        self.T = T

    @staticmethod
    def from_labels(labels, t_per_anchor, fraction):
        lab = labels.detach()
        counts = torch.bincount(lab.reshape(-1).to(torch.int64)).cpu()      # host sync unless labels are on the CPU
        return TripletPlan(counts, t_per_anchor, fraction)

    def host_draws(self):
        """Consumes the CPU default generator exactly like the reference."""
        pos, neg = [], []
        for n_a, n_neg, T_c in self.draw_specs:
            pos.append(torch.randint(0, n_a - 1, (T_c,)))
            neg.append(torch.randint(0, n_neg, (T_c,)))
        if not pos:
            e = torch.empty(0, dtype=torch.int64)
            return e, e.clone()
        return torch.cat(pos), torch.cat(neg)


def class_index(labels, C):
    """labels int64 [M] on the device -> (order int32 [M]: members grouped by class in ascending index,
    offsets int32 [C+1])."""
    dev = labels.device
    M = labels.numel()
    order = torch.empty(M, dtype=torch.int32, device=dev)
    offsets = torch.empty(C + 1, dtype=torch.int32, device=dev)
    nbytes = _lib.call("hpcs_class_index_workspace", M, C)
    counts_ws = torch.empty(max(nbytes // 4, 1), dtype=torch.int32, device=dev)
    _lib.call("hpcs_class_index", labels, M, C, order, offsets, counts_ws, nbytes)
    return order, offsets


def sample_with_plan(labels, plan, rng="host", seed=0):
    """labels: int64 [M] on the device.  Returns (a, p, n) int64 [T] on the device."""
    dev = labels.device
    if not labels.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")
    labels = labels.reshape(-1).to(torch.int64).contiguous()
    M, C, T = labels.numel(), plan.C, plan.T
    a = torch.empty(T, dtype=torch.int64, device=dev)
    p = torch.empty(T, dtype=torch.int64, device=dev)
    n = torch.empty(T, dtype=torch.int64, device=dev)
    if T == 0:
        return a, p, n
    order, offsets = class_index(labels, C)
    from .._pinned import upload
    toff = upload(plan.toff, dev, key="plan")
    kk = upload(plan.k, dev, key="plan")
    dp = dn = None
    if rng == "host":
        hp, hn = plan.host_draws()
        dp, dn = upload(hp, dev, key="draws"), upload(hn, dev, key="draws")
    elif rng != "device":
        raise ValueError("rng must be 'host' or 'device'")
    _lib.call("hpcs_sample_triplets", order, offsets, toff, kk, C, M, T, dp, dn, int(seed) & (2 ** 63 - 1), a, p, n)
    return a, p, n


def get_balanced_random_triplet_indices(labels, ref_labels=None, t_per_anchor=None, fraction=None, weights=None,
                                        rng="host", seed=0):
    if ref_labels is not None and ref_labels is not labels:
        raise NotImplementedError("ref_labels != labels is never used by HPCS (triplet_margin_miner.py:14-15)")
    if weights is not None:
        raise NotImplementedError("weighted negative sampling is never used by HPCS")
    plan = TripletPlan.from_labels(labels, t_per_anchor, fraction)
    return sample_with_plan(labels, plan, rng=rng, seed=seed)
