"""Mirror of hpcs/loss/ultrametric_loss.py:16-143 (MetricHyperbolicLoss) on HIP kernels.

compute_loss(x_euclidean, x_poincare, labels) -> {"loss_hyp": {"losses": t}, "loss_metric": {"losses": t}}.

What differs in HOW (DESIGN.md "Loss"):
  * the (B*N)^2 cosine matrix is never built: the miner and the HypHC term gather row dots, and
    mat_sim.mean() is evaluated in closed form 0.5 + 0.5*|sum_i x_hat_i|^2 / M^2;
  * the LCA distance of two points on the same diameter-scaled sphere is evaluated from their similarity in
    closed form (fp64 scalars) instead of the reference's cancellation-bound fp32 reflections;
  * the number of mined triplets stays on the device (no host sync inside the step).
PML (absent) behaviour restated here: CosFaceLoss = normalised-weights softmax with cos-margin 0.35, scale 2,
mean reduction; parity unpinned.
"""
from typing import Union

import torch

from .. import _lib
from ..distances import CosineSimilarity
from ..miner.loss_and_miner_utils import TripletPlan
from ..miner.triplet_margin_loss import TripletMarginLoss
from ..miner.triplet_margin_miner import RandomTripletMarginMiner

call = _lib.call


class L2RowsFn(torch.autograd.Function):
    """x [M,E] -> x / max(|x|, 1e-12)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        xn = torch.empty_like(x)
        inv = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        call("hpcs_l2_rows_fwd", x, x.shape[0], x.shape[1], xn, inv)
        ctx.save_for_backward(xn, inv)
        return xn

    @staticmethod
    def backward(ctx, g):
        xn, inv = ctx.saved_tensors
        gx = torch.empty_like(xn)
        call("hpcs_l2_rows_bwd", xn, inv, g.contiguous().float(), xn.shape[0], xn.shape[1], gx, 0)
        return gx


_KIND = {None: -1, "all": 0, "easy": 1, "semihard": 2, "hard": 3}
_LAST_KEPT = [None]          # device int64 tensor: number of triplets the last HypHC forward averaged over


class HypHCFn(torch.autograd.Function):
    """mean over mined triplets of sum(w) - sum(w*softmax(d/temp)), plus the closed-form mean similarity.

    kind=None: (a,p,n,count) are already mined.  kind in {"all","easy","semihard","hard"}: (a,p,n) are the sampled
    triplets and the miner's margin rule is evaluated inside the loss kernels (no compaction pass).
    When a gradient is needed the forward runs the gradient kernel in one-pass mode (loss + kept count + unscaled
    gradient in a single sweep over the triplets); backward only rescales by upstream/kept.
    """

    @staticmethod
    def forward(ctx, xn, scale, a, p, n, count, temperature, margin, kind):
        M, E = xn.shape
        dev = xn.device
        loss_sum = torch.empty(1, dtype=torch.float64, device=dev)
        kid = _KIND[kind]
        kept = torch.empty(1, dtype=torch.int64, device=dev)
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gxn_raw = gscale_raw = None
        if need_grad:
            ones_f = torch.ones(1, device=dev)
            ones_i = torch.ones(1, dtype=torch.int64, device=dev)
            gxn_raw = torch.zeros(M, E, device=dev)
            gscale_raw = torch.empty(1, dtype=torch.float64, device=dev)
            call("hpcs_hyphc_bwd", xn, E, a, p, n, count, a.numel(), scale, temperature, margin, kid, ones_i, ones_f,
                 gxn_raw, gscale_raw, loss_sum, kept)
        else:
            call("hpcs_hyphc_fwd", xn, E, a, p, n, count, a.numel(), scale, temperature, margin, kid, loss_sum,
                 kept if kid >= 0 else None)
        norm = kept if (kid >= 0 or need_grad) else count
        if kid < 0 and need_grad:
            norm = count                       # pre-mined triplets: the mean is over `count`
        _LAST_KEPT[0] = norm
        colsum = torch.empty(E, dtype=torch.float64, device=dev)
        call("hpcs_colsum", xn, M, E, colsum)
        mean_sim = 0.5 + 0.5 * (colsum * colsum).sum() / float(M * M)
        out = loss_sum[0] / norm[0].double() + mean_sim
        ctx.save_for_backward(scale, colsum, norm, gxn_raw, gscale_raw)
        ctx.M = M
        return out.float()

    @staticmethod
    def backward(ctx, g):
        scale, colsum, norm, gxn_raw, gscale_raw = ctx.saved_tensors
        M = ctx.M
        g32 = g.reshape(1).float()
        inv = (g32.double() / norm.double().clamp_min(1)).float()          # upstream / kept
        # d mean_sim / d xn_i = colsum / M^2 (same row for every i)
        gxn = torch.addcmul((colsum / float(M * M)).float().mul(g32).unsqueeze(0), gxn_raw, inv)
        gscale = (gscale_raw * inv.double()).float().view_as(scale)
        return gxn, gscale, None, None, None, None, None, None, None


def sampled_width(E):
    """Row width hpcs_hyphc_sampled runs an embedding size on (16 / 32 / 64), or None beyond 64."""
    return 16 if E <= 16 else 32 if E <= 32 else 64 if E <= 64 else None


class HypHCSampledFn(torch.autograd.Function):
    """HypHC term with the class-balanced triplets generated INSIDE the kernel (device RNG): no index arrays, no
    sampler / filter passes; loss, kept count and unscaled gradient in one sweep (hpcs_hyphc_sampled)."""

    @staticmethod
    def forward(ctx, xn, scale, labels, order, offsets, toff, kk, T, seed, temperature, margin, kind, debug, seed_dev=None):
        """T sizes the grid (>= toff[C], the count the kernel reads on the device); seed_dev (int64[1] device tensor)
        overrides the by-value seed -- both so that a captured launch can be replayed with per-step values."""
        M, E_in = xn.shape
        dev = xn.device
        C = kk.numel()
        # The kernel gathers rows as 16-byte pieces, 8 lanes per row: widths 16 / 32 / 64.  Any other embedding size (the
        # reference shipped 50, 4 and 15: checkpoints/*/config.yaml) runs on a zero-padded copy -- zero columns change no dot
        # product and their gradient is exactly zero; backward returns the first E_in columns.
        E = sampled_width(E_in)
        if E != E_in:
            xk = torch.zeros(M, E, device=dev)
            xk[:, :E_in] = xn
            xn = xk
        loss_sum = torch.empty(1, dtype=torch.float64, device=dev)
        kept = torch.empty(1, dtype=torch.int64, device=dev)
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gxn_raw = torch.empty(M, E, device=dev) if need_grad else None          # zero-filled by the entry point's own zero launch
        gscale_raw = torch.empty(1, dtype=torch.float64, device=dev) if need_grad else None
        dbg = [None, None, None]
        if debug is not None:
            dbg = [torch.zeros(T, dtype=torch.int64, device=dev) for _ in range(3)]
        call("hpcs_hyphc_sampled", xn, labels, order, offsets, toff, kk, C, M, E, T, int(seed) & (2 ** 63 - 1), seed_dev,
             scale, temperature, margin, _KIND[kind], loss_sum, kept, gxn_raw, gscale_raw, dbg[0], dbg[1], dbg[2])
        if debug is not None:
            # (a, p, n, kept): a triplet the miner's rule dropped is exported with ~anchor (negative)
            live = dbg[0] >= 0
            debug.extend([torch.where(live, dbg[0], -1 - dbg[0]), dbg[1], dbg[2], live])
        _LAST_KEPT[0] = kept
        ctx.E_in = E_in
        colsum = torch.empty(E, dtype=torch.float64, device=dev)
        call("hpcs_colsum", xn, M, E, colsum)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        call("hpcs_hyphc_finalize", loss_sum, kept, colsum, M, E, out)        # loss_sum / kept + 0.5 + 0.5 |colsum|^2 / M^2
        ctx.save_for_backward(scale, colsum, kept, gxn_raw, gscale_raw)
        ctx.M = M
        return out[0]

    @staticmethod
    def backward(ctx, g):
        scale, colsum, kept, gxn_raw, gscale_raw = ctx.saved_tensors
        M = ctx.M
        g32 = g.reshape(1)
        if g32.dtype != torch.float32:
            g32 = g32.float()
        gscale = torch.empty(1, dtype=torch.float32, device=gxn_raw.device)
        # in place on the kernel's unscaled gradient (a saved tensor nobody reads again): one launch
        call("hpcs_hyphc_grad_finalize", g32, kept, colsum, gscale_raw, M, gxn_raw.shape[1], gxn_raw, gscale)
        if ctx.E_in != gxn_raw.shape[1]:
            gxn_raw = gxn_raw[:, :ctx.E_in].contiguous()
        return (gxn_raw, gscale.view_as(scale)) + (None,) * 12


_ONES = {}


def _ones1(device):
    """A persistent device scalar 1.0 (the upstream gradient of the unscaled backward passes)."""
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones(1, device=device)
    return t


class CosFaceFn(torch.autograd.Function):
    """CosFace / hierarchical CosFace mean loss.  The loss is linear in its upstream gradient, so when a gradient will be
    needed the forward also runs the backward kernel with upstream 1 (like the sampled HypHC kernel does) and backward is one
    rescale of that buffer.  Inside the step driver's forward_side_stream() all of it is enqueued on the SIDE stream: nothing on
    the main stream needs the CosFace branch until the two loss terms are added, so it runs beside the sampled HypHC kernel
    (which is alone on the chip for 1.7 ms) instead of in the launch-latency-bound tail behind it -- the caller joins with
    ops.join_forward_side() after the HypHC term has been enqueued."""

    @staticmethod
    def forward(ctx, xn, wn, labels, margin, scale, branch_id):
        from .. import ops
        M, E = xn.shape
        C = wn.shape[0]
        dev = xn.device
        n_levels = 1 if branch_id is None else branch_id.shape[0]
        logits = torch.empty(M, C, device=dev, dtype=torch.float32)
        rows = torch.empty(M, device=dev, dtype=torch.float32)
        lse = torch.empty(M, device=dev, dtype=torch.float32)
        mean = torch.empty(1, device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gbuf = torch.empty(M * E + C * E, device=dev, dtype=torch.float32) if need else None     # [gxn | gwn], upstream 1
        ones = _ones1(dev) if need else None
        side = ops.fork_forward_side()
        with (torch.cuda.stream(side) if side is not None else _NullCtx()):
            call("hpcs_cosface_fwd", xn, wn, labels, branch_id, n_levels, M, E, C, margin, scale, logits, rows, lse)
            call("hpcs_rowsum", rows, 1, M, 1.0 / M, mean)
            if need:
                call("hpcs_cosface_bwd", xn, wn, labels, logits, lse, branch_id, n_levels, M, E, C, scale, ones,
                     gbuf[:M * E], gbuf[M * E:])
        if side is not None:
            # every buffer a side-stream kernel touches stays referenced until the step's streams have joined (the end of the
            # backward pass: ops.deferred_weight_grads): a temporary freed here would go back to the MAIN stream's pool while
            # the side stream still writes it -- `rows` [M] handed out again as the miner's class index was a GPU memory fault
            ops._SideWGrad.keep.append((logits, rows, lse, mean, gbuf, ones, xn, wn, labels, branch_id))
        ctx.save_for_backward(gbuf)
        ctx.dims = (M, E, C)
        ctx.mark_non_differentiable(logits)
        return mean[0], logits

    @staticmethod
    def backward(ctx, g, _glogits):
        (gbuf,) = ctx.saved_tensors
        M, E, C = ctx.dims
        gb = gbuf * g.reshape(1).float()
        return gb[:M * E].view(M, E), gb[M * E:].view(C, E), None, None, None, None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class CosFaceLoss(torch.nn.Module):
    """PML CosFaceLoss(num_classes, embedding_size, margin, scale): W [E,C] ~ N(0,1), mean cross entropy."""

    def __init__(self, num_classes, embedding_size, margin=0.35, scale=64):
        super().__init__()
        self.num_classes = num_classes
        self.embedding_size = embedding_size
        self.margin = margin
        self.scale = scale
        self.W = torch.nn.Parameter(torch.Tensor(embedding_size, num_classes))
        torch.nn.init.normal_(self.W)
        self.branch_id = None

    def logits_and_loss(self, embeddings, labels, xn=None):
        if xn is None:
            xn = L2RowsFn.apply(embeddings)
        wn = L2RowsFn.apply(self.W.t())
        bid = self.branch_id
        if bid is not None and bid.device != xn.device:
            bid = self.branch_id = bid.to(xn.device)
        return CosFaceFn.apply(xn, wn, labels.reshape(-1).to(torch.int64).contiguous(), float(self.margin),
                               float(self.scale), bid)

    def forward(self, embeddings, labels, xn=None):
        loss, logits = self.logits_and_loss(embeddings, labels, xn)
        self.last_logits = logits               # margin-adjusted, scaled: what get_logits() returns (in-step metrics)
        return loss


def class_metrics(logits, labels):
    """(accuracy, IoU) device scalars from [M,C] logits and int labels: base_hyp_hc.py:88-99 without torchmetrics
    (micro top-1 accuracy; macro Jaccard over the classes present in predictions or targets)."""
    logits = logits.detach().contiguous().float()
    M, C = logits.shape
    counts = torch.empty(3 * C, dtype=torch.int64, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    call("hpcs_class_metrics", logits, labels.reshape(-1).to(torch.int64).contiguous(), M, C, counts, out)
    if _lib.WATCH is not None and not _lib.WATCH.armed:          # diagnostics: watch the allocator block holding `out`
        _lib.WATCH.arm(out.data_ptr() & ~511)
    return out[0], out[1]


def branch_ids(hierarchy_list, num_classes):
    """[level][branch][class ids] (hpcs/data/hierarchy_list.py:9-19) -> int32 [levels, C]: the branch of every class per
    level; a class that no branch of a level lists keeps a private id (>= C), i.e. its own probability -- what the
    reference's in-place slice assignment leaves there (hierarchical_cosface_loss.py:13-24)."""
    if not 1 <= len(hierarchy_list) <= 8:
        raise ValueError("hierarchy_list must hold 1..8 levels")
    bid = torch.arange(num_classes, dtype=torch.int32).repeat(len(hierarchy_list), 1) + num_classes
    for lv, level in enumerate(hierarchy_list):
        for j, branch in enumerate(level):
            for ch in branch:
                bid[lv, ch] = j
    return bid.contiguous()


class HierarchicalCosFaceLoss(CosFaceLoss):
    """hpcs/loss/hierarchical_cosface_loss.py:31-87: CosFace logits -> softmax -> per hierarchy level the class
    probabilities of a branch are replaced by the branch sum -> NLL; levels are summed.
    hierarchy_list: [level][branch][class ids] (hpcs/data/hierarchy_list.py:9-19)."""

    def __init__(self, num_classes, embedding_size, margin=0.35, scale=64, hierarchy_list=None):
        super().__init__(num_classes, embedding_size, margin=margin, scale=scale)
        self.hierarchy_list = hierarchy_list or []
        self.branch_id = branch_ids(self.hierarchy_list, num_classes)

    def compute_loss(self, embeddings, labels, indices_tuple=None, ref_emb=None, ref_labels=None):
        """PML loss-dict contract (hierarchical_cosface_loss.py:57-87): the level-summed NLL is a scalar that the
        reference multiplies by per-element miner weights (all ones without a miner) -> 'element' losses [M]."""
        if indices_tuple is not None or ref_emb is not None or ref_labels is not None:
            raise NotImplementedError("miner weights / reference embeddings are never passed by HPCS (ultrametric_loss.py:165)")
        loss = self.forward(embeddings, labels)
        M = embeddings.shape[0]
        return {"loss": {"losses": loss.expand(M), "indices": torch.arange(M, device=embeddings.device),
                         "reduction_type": "element"}}


class MetricHyperbolicLoss(torch.nn.Module):
    def __init__(self, margin: float = 1.0, t_per_anchor: int = 50, fraction: float = 1.2,
                 scale: Union[float, torch.Tensor, torch.nn.Parameter] = 1e-3, temperature: float = 0.05,
                 anneal_factor: float = 0.5, num_class: int = 4, embedding_size: int = 4, cosface: bool = True,
                 miner: bool = False, rng: str = "host", fused_filter: bool = False):
        super().__init__()
        self.margin = margin
        self.t_per_anchor = t_per_anchor
        self.fraction = fraction
        self.scale = scale
        self.temperature = temperature
        self.anneal_factor = anneal_factor
        self.num_class = num_class
        self.embedding_size = embedding_size
        self.cosface = cosface
        self.miner = miner
        self.fused_filter = fused_filter        # True: hyp_miner's margin filter runs inside the loss kernels
        self.distance_sim = CosineSimilarity()
        if self.miner:
            self.hyp_miner = RandomTripletMarginMiner(distance=self.distance_sim, margin=0, t_per_anchor=t_per_anchor,
                                                      fraction=fraction, type_of_triplets="easy", rng=rng)
        if self.cosface:
            self.loss_cosface = CosFaceLoss(num_classes=num_class, embedding_size=embedding_size, margin=0.35, scale=2)
        else:
            self.triplet_miner = RandomTripletMarginMiner(distance=self.distance_sim, margin=margin,
                                                          t_per_anchor=t_per_anchor, fraction=fraction,
                                                          type_of_triplets="semihard", rng=rng)
            self.loss_triplet = TripletMarginLoss(distance=self.distance_sim, margin=margin)
        self.last_triplets = None
        self.last_T = 0
        self.debug_triplets = None      # tests: set to a list to receive the triplets generated in-kernel

    # ---- pieces ------------------------------------------------------------------------------------------
    def _scale_tensor(self, like):
        s = self.scale
        if not torch.is_tensor(s):
            s = torch.tensor([float(s)], device=like.device)
        return s.reshape(1) if s.numel() == 1 else s

    def get_triplets(self, n_samples, device=None):
        """Miner-less fallback (ultrametric_loss.py:42-55): all pairs x t_per_anchor random thirds; index bookkeeping on
        the host with the reference's CPU draw order (one torch.randint of n_pairs*t_per_anchor), like the reference.
        device=None: the reference's return (a, p, n) on the CPU; else (a, p, n, count) on `device` for HypHCFn."""
        ij = torch.combinations(torch.arange(n_samples), r=2).repeat_interleave(self.t_per_anchor, dim=0)
        k = torch.randint(n_samples, (ij.shape[0],), dtype=torch.long)
        keep = (ij[:, 0] != k) & (ij[:, 1] != k)
        if device is None:
            return ij[keep, 0], ij[keep, 1], k[keep]
        a, p, n = ij[keep, 0].to(device), ij[keep, 1].to(device), k[keep].to(device)
        return a, p, n, torch.full((1,), a.numel(), dtype=torch.int64, device=device)

    def compute_hyp(self, x_poincare, labels, plan=None, triplets=None, xn=None, step_ctx=None):
        """step_ctx (hpcs_amd.step.StepCtx, optional): the step driver has already drawn the miner's seed and uploaded the
        plan into the per-step parameter block; the kernels read both from there."""
        if xn is None:
            xn = L2RowsFn.apply(x_poincare)
        kind, margin = None, 0.0
        if triplets is None:
            if self.miner and self.fused_filter:
                # sample only; the 'easy'/margin-0 rule of hyp_miner is evaluated inside the loss kernels
                from ..miner.loss_and_miner_utils import TripletPlan, class_index, sample_with_plan
                from .._pinned import upload
                m = self.hyp_miner
                if plan is None:
                    plan = TripletPlan.from_labels(labels, m.t_per_anchor, m.fraction)
                from_ctx = step_ctx is not None and step_ctx.plan is plan and step_ctx.miner_seed is not None
                seed = step_ctx.miner_seed if from_ctx else m.next_seed()
                if m.rng == "device" and sampled_width(xn.shape[1]) is not None and plan.T > 0:
                    # device RNG: triplets are generated inside the loss kernel (no index arrays at all)
                    lab = labels.reshape(-1).to(torch.int64).contiguous()
                    order, offsets = class_index(lab, plan.C)
                    if from_ctx:
                        toff, kk, seed_dev, T_grid = step_ctx.toff_dev, step_ctx.kk_dev, step_ctx.seed_dev["miner"], step_ctx.T_grid
                    else:
                        toff = upload(plan.toff, xn.device, key="plan")
                        kk = upload(plan.k, xn.device, key="plan")
                        seed_dev, T_grid = None, plan.T
                    self.last_triplets = None
                    self.last_T = plan.T
                    out = HypHCSampledFn.apply(xn, self._scale_tensor(xn), lab, order, offsets, toff, kk, T_grid, seed,
                                               float(self.temperature), float(m.margin), m.type_of_triplets,
                                               self.debug_triplets, seed_dev)
                    self.last_kept = _LAST_KEPT[0]
                    return out
                a, p, n = sample_with_plan(labels, plan, rng=m.rng, seed=seed)
                triplets = (a, p, n, torch.full((1,), a.numel(), dtype=torch.int64, device=xn.device))
                kind, margin = m.type_of_triplets, float(m.margin)
            elif self.miner:
                triplets = self.hyp_miner.mine_device(x_poincare, labels, plan=plan, xn=xn.detach())
            else:
                triplets = self.get_triplets(x_poincare.shape[0], x_poincare.device)
        a, p, n, count = triplets
        self.last_triplets = triplets
        self.last_T = a.numel()
        scale = self._scale_tensor(xn)
        out = HypHCFn.apply(xn, scale, a, p, n, count, float(self.temperature), margin, kind)
        self.last_kept = _LAST_KEPT[0]
        return out

    def get_logits(self, embeddings, labels):
        if not hasattr(self, "loss_cosface"):
            raise ValueError("Cannot get logits since this class doesn't use any CosFaceLoss")
        with torch.no_grad():
            return self.loss_cosface.logits_and_loss(embeddings, labels)[1]

    def compute_loss(self, x_euclidean, x_poincare, labels, *args, plan=None, plan_metric=None, step_ctx=None):
        if not x_poincare.is_cuda:
            raise _lib.HpcsNativeError("hpcs_amd losses need HIP-device tensors (there is no CPU fallback)")
        labels = labels.reshape(-1)
        xn = L2RowsFn.apply(x_poincare)
        if self.cosface:
            # enqueued first: inside the step driver its kernels go to the side stream and run beside the HypHC kernel below
            from ..ops import join_forward_side
            loss_metric = self.loss_cosface(x_poincare, labels.long(), xn=xn)
            try:
                loss_hyperbolic = self.compute_hyp(x_poincare, labels, plan=plan, xn=xn, step_ctx=step_ctx)
            finally:
                join_forward_side()
            return {"loss_hyp": {"losses": loss_hyperbolic}, "loss_metric": {"losses": loss_metric}}
        loss_hyperbolic = self.compute_hyp(x_poincare, labels, plan=plan, xn=xn, step_ctx=step_ctx)
        a, p, n, count = self.triplet_miner.mine_device(x_poincare, labels, plan=plan_metric or plan, xn=xn.detach())
        loss_metric = self.loss_triplet(x_poincare, labels, (a, p, n), count=count)
        return {"loss_hyp": {"losses": loss_hyperbolic}, "loss_metric": {"losses": loss_metric}}

    def make_plan(self, labels_cpu_or_counts, minlength=0):
        """Per-class sampling plan from host-side labels (call before the forward pass: no mid-step sync).
        minlength: pad the class table (absent classes sample nothing), so that its size does not depend on the batch."""
        t = labels_cpu_or_counts
        counts = torch.bincount(t.reshape(-1).to(torch.int64).cpu(), minlength=minlength)
        return TripletPlan(counts, self.t_per_anchor, self.fraction)

    def anneal_temperature(self):
        """temperature *= clamp(anneal_factor, 0.2, 1) (ultrametric_loss.py:133-137).  A TENSOR anneal_factor behaves exactly
        like the reference (the temperature becomes a tensor; pinned by tests/golden/behaviour_r3.json).  DEVIATION, stated
        in INTEGRATION.md: with a Python number -- the reference's own default, 0.5 -- the reference raises TypeError
        (torch.clamp on a float), so `--anneal-step > 0` cannot run there; here the clamp is evaluated on the number."""
        f = self.anneal_factor
        if torch.is_tensor(f):
            self.temperature = self.temperature * torch.clamp(f, 0.2, 1)
        else:
            self.temperature *= min(max(float(f), 0.2), 1.0)
        return self.temperature

    def normalize_embeddings(self, embeddings):
        """F.normalize(embeddings, p=2, dim=1) * clamp(scale, 1e-4, 1) (ultrametric_loss.py:139-143); the reference only
        ever passes [T, E] rows (:75-77)."""
        if embeddings.dim() != 2:
            raise ValueError("normalize_embeddings expects [T, E] rows (ultrametric_loss.py:75-77)")
        return self.normalize_rows(embeddings)

    def normalize_rows(self, embeddings):
        """normalize_embeddings over the LAST dim of [..., E] (the batched decode path: one call for [B, N, E])."""
        s = self._scale_tensor(embeddings)
        rows = embeddings.reshape(-1, embeddings.shape[-1])
        return (L2RowsFn.apply(rows) * torch.clamp(s, 1e-4, 1)).reshape(embeddings.shape)


class HierarchicalMetricHyperbolicLoss(MetricHyperbolicLoss):
    """hpcs/loss/ultrametric_loss.py:146-176: the PartNet default -- HypHC term + hierarchical CosFace."""

    def __init__(self, margin: float = 1.0, t_per_anchor: int = 50, fraction: float = 1.2,
                 scale: Union[float, torch.Tensor, torch.nn.Parameter] = 1e-3, temperature: float = 0.05,
                 anneal_factor: float = 0.5, num_class: int = 4, embedding_size: int = 4, miner: bool = False,
                 hierarchy_list: list = [], rng: str = "host", fused_filter: bool = False):
        super().__init__(margin=margin, t_per_anchor=t_per_anchor, fraction=fraction, scale=scale,
                         temperature=temperature, anneal_factor=anneal_factor, num_class=num_class,
                         embedding_size=embedding_size, cosface=True, miner=miner, rng=rng, fused_filter=fused_filter)
        self.hierarchy_list = hierarchy_list
        self.loss_cosface = HierarchicalCosFaceLoss(num_classes=num_class, embedding_size=embedding_size, margin=0.35,
                                                    scale=2, hierarchy_list=hierarchy_list)
