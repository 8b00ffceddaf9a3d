"""Training-step driver: the ordering of BaseSimilarityHypHC.forward / training_step + RAdam.step
(hpcs/models/base_hyp_hc.py:101-171, hpcs/models/shapenet_hyp_hc.py:55-91) without Lightning.

The reference's LightningModule is the CALLER of the hot path and stays out of scope; this
module is the minimal stand-in that the benchmark, the smoke test and the data-parallel wrapper drive:

    rotate (SO(3), on device) -> nn_feat -> nn_emb -> flatten -> MetricHyperbolicLoss.compute_loss
    -> total = loss_metric + trade_off*loss_hyp -> backward -> [all-reduce] -> RAdam.step
"""
import torch

from .dp import FlatGradBucket
from .loss.ultrametric_loss import HierarchicalMetricHyperbolicLoss, MetricHyperbolicLoss
from .ops import PrepCache, deferred_weight_grads, forward_side_stream, zero_arena_release, zero_arena_reset
from .optim import RAdam

FORWARD_SIDE = __import__("os").environ.get("HPCS_FWD_SIDE", "1") != "0"      # A/B: graph transposes in the forward pass


_GC_SETTLED = False


def _gc_settle():
    """Once per process, after the first few training steps: one full collection, then gc.freeze().  CPython's first
    generation-2 collection after start-up walks every object the imports created (torch + numpy: ~270 k containers here,
    40-100 ms measured) and lands within the first ~50 steps of a loop -- a host stall of 4-9 step times with the GPU idle
    (r03: it was the 10.9 ... 16 ms run-to-run spread of 20-step measurements).  Freezing moves the objects alive now into
    the permanent generation: later full collections only walk what training allocated since.  Collection itself stays on.
    OPT-IN (a library must not freeze the embedding application's heap behind its back): HPCS_GC_FREEZE=1, which
    bench.py sets for its own process; INTEGRATION.md tells a training script how to do the same."""
    global _GC_SETTLED
    if _GC_SETTLED:
        return
    _GC_SETTLED = True
    if __import__("os").environ.get("HPCS_GC_FREEZE", "0") == "1":
        import gc
        gc.collect()
        gc.freeze()


def _new_graph():
    """A CUDAGraph that keeps its hipGraph_t so that the capture can be inspected (torch >= 2.8: keep_graph)."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True), True
    except TypeError:
        return torch.cuda.CUDAGraph(), False


def _lib_graph_census(graph):
    from . import _lib
    return _lib.graph_census(graph)


def _watch_disarm(freeze=False):
    """Diagnostics (hpcs_amd/_lib.py Watch): the watch armed by class_metrics ends with the step; a capture's call list is
    kept (`freeze`) because replays execute exactly that list."""
    from . import _lib
    w = _lib.WATCH
    if w is not None and w.armed:
        w.disarm()
        if freeze:
            w.frozen = True


def random_rotations(B, device, generator=None):
    """Haar-uniform SO(3) matrices [B,3,3] from normalised Gaussian quaternions, generated on the device
    (no host round trip: the reference's CPU rotation, shapenet_hyp_hc.py:63-69, is the step's only other host hop)."""
    q = torch.randn(B, 4, device=device, generator=generator)
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(B, 3, 3)


def rotate_to_channels_first(points, rot=None, seed=0, return_rot=False, seed_dev=None):
    """points [B,N,3] -> [B,3,N] = transpose(points @ R) with one Haar rotation per cloud, in ONE launch
    (shapenet_hyp_hc.py:63-73).  rot [B,3,3] given: use it; else quaternions from the counter RNG keyed by `seed`
    (or by the device word seed_dev, int64[1], when given)."""
    from . import _lib
    if not points.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd needs HIP-device tensors (no CPU fallback)")
    pts = points.contiguous() if points.dtype == torch.float32 else points.float().contiguous()
    B, N, _ = pts.shape
    x = torch.empty(B, 3, N, device=pts.device, dtype=torch.float32)
    R = torch.empty(B, 3, 3, device=pts.device, dtype=torch.float32) if return_rot else None
    rin = None if rot is None else rot.to(torch.float32).contiguous()
    _lib.call("hpcs_so3_rotate_points", pts, B, N, int(seed) & (2 ** 63 - 1), seed_dev, rin, x, R)
    return (x, R) if return_rot else x


from .utils.data import to_categorical  # noqa: E402,F401  (hpcs/utils/data.py:24-29, on the device of y)


class StepCtx:
    """What the host decided for ONE step before any kernel of it is launched: seeds, the miner's plan, RAdam's tables --
    already copied into the per-step parameter block (hpcs_amd/_params.py); the fields are the by-value copies and the
    persistent device views the kernels read."""
    __slots__ = ("plan", "T_grid", "miner_seed", "rot_seed", "drop_seeds", "seed_dev", "toff_dev", "kk_dev",
                 "radam_tables", "radam_dev")


class HypHCStep(torch.nn.Module):
    def __init__(self, nn_feat, nn_emb, hyp_size, num_class, num_categories=16, lr=1e-3, margin=0.5, t_per_anchor=50,
                 fraction=1.2, temperature=0.05, anneal_factor=0.5, trade_off=0.1, miner=True, cosface=True,
                 train_rotation="so3", rng="device", fused_filter=True, hierarchical=False, hierarchy_list=None,
                 test_rotation="so3"):
        super().__init__()
        self.nn_feat = nn_feat
        self.nn_emb = nn_emb
        self.num_categories = num_categories
        self.trade_off = trade_off
        self.train_rotation = train_rotation
        self.test_rotation = test_rotation
        self.scale = torch.nn.Parameter(torch.Tensor([1e-3]), requires_grad=True)        # base_hyp_hc.py:62
        common = dict(margin=margin, t_per_anchor=t_per_anchor, fraction=fraction, scale=self.scale,
                      temperature=temperature, anneal_factor=anneal_factor, num_class=num_class, embedding_size=hyp_size,
                      miner=miner, rng=rng, fused_filter=fused_filter)
        if hierarchical:                       # PartNet default (partnet_hyp_hc.py:57-67)
            self.metric_hyp_loss = HierarchicalMetricHyperbolicLoss(hierarchy_list=hierarchy_list, **common)
        else:
            self.metric_hyp_loss = MetricHyperbolicLoss(cosface=cosface, **common)
        self.lr = lr
        self._rot_seed = None
        self._rot_calls = 0
        self.side_stream_wgrad = __import__("os").environ.get("HPCS_SIDE_STREAM", "1") != "0"   # 0: serial (profiling)
        # capture form of the replayed step: "side" (fork / join over the weight-gradient stream) or "serial" (one stream);
        # auto mode measures both, HPCS_GRAPH=1 takes "side" unless HPCS_GRAPH_FORM / HPCS_SIDE_STREAM=0 say otherwise
        forced_form = __import__("os").environ.get("HPCS_GRAPH_FORM", "").lower()
        self._graph_form = forced_form if forced_form in ("side", "serial") else ("side" if self.side_stream_wgrad else "serial")
        self._off_shape, self._off_shape_key = 0, None
        # hipGraph replay of the steady-state step: HPCS_GRAPH = 1 (always), 0 (never) or auto (default: after the warm-up
        # the step is captured, two eager steps and two replays are timed with events, and the faster mode is kept --
        # replay wins when the step is launch-bound (small clouds: 2x at B=4, N=256), eager when it is GPU-bound and the
        # side stream overlaps better than the graph's own scheduling does (B=32, N=1024: replay measured 3.7 % slower)).
        # Eligible steps: device RNG, miner + fused filter, host copy of the labels given, no caller-supplied rotation /
        # dropout masks.
        mode = __import__("os").environ.get("HPCS_GRAPH", "auto").lower()
        self.use_graph = mode not in ("0", "off", "false")
        self.graph_auto = mode == "auto"
        self.graph_warmup = 3               # eager steps before the capture (allocator pools, lazily built tables)
        self._graph = None
        self._prep_cache = PrepCache()                 # rebuilt in configure() once the parameters are on the device
        self._eager_steps = 0
        self._calls = 0
        self._tune = None                   # auto mode: {"phase", "t": {"eager": [...], "graph": [...]}, "ev"}
        self.metrics = True
        self.optimizer = None
        self.bucket = None
        self.params_block = None
        self.num_class = num_class
        self.poisoned = False               # True: a step raised after its optimiser launch (see _rewind_or_poison)
        self.last_capture_nodes = None      # node census of the last hipGraph capture ({"fwd_bwd": {...}, "opt": {...}})

    def configure(self):
        """Call once the module is on its device: flat gradient bucket + RAdam over ALL parameters (base_hyp_hc.py:143)
        + the per-step parameter block (seeds, miner plan, RAdam tables: one upload per step)."""
        from ._params import StepParams
        params = list(self.parameters())
        pairs = [(m.map_to_feat.weight, m.map_to_dir.weight) for m in self.modules()
                 if hasattr(m, "map_to_feat") and hasattr(m, "map_to_dir") and hasattr(m, "batchnorm")]
        self.bucket = FlatGradBucket(params, adjacent=pairs)
        self.optimizer = RAdam(params, lr=self.lr)
        group = self.optimizer.param_groups[0]
        step0 = group.get("step", 0)
        tables = self.optimizer.build_tables()            # only to size the block: undo its step-count advance
        group["step"] = step0
        self.params_block = StepParams(params[0].device, self.num_class, [t[1].numel() for t in tables])
        self._prep_cache = PrepCache(params)
        return self

    def _next_rot_seed(self):
        from .miner.triplet_margin_miner import mix64
        if self._rot_seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self._rot_seed = mix64((int(torch.initial_seed()) & (2 ** 64 - 1)) ^ mix64(0xE7037ED1A0B428DB * (rank + 1) & (2 ** 64 - 1)))
        self._rot_calls += 1            # counter RNG stream: (torch seed at first use, rank, call index), mixed
        return mix64(self._rot_seed ^ mix64(self._rot_calls)) & (2 ** 63 - 1)

    def _prepare(self, targets_host, own_rotation, own_dropout, T_cap=0):
        """Host half of a step (before any kernel is launched): plan, seeds, RAdam tables -> parameter block -> ONE
        asynchronous copy.  Returns the StepCtx the kernels' launchers read."""
        loss, blk = self.metric_hyp_loss, self.params_block
        ctx = StepCtx()
        ctx.plan = None
        ctx.miner_seed = None
        ctx.T_grid = 0
        if targets_host is not None and loss.miner:
            ctx.plan = loss.make_plan(targets_host, minlength=self.num_class)
            if ctx.plan.C != self.num_class:
                raise ValueError(f"labels reach class {ctx.plan.C - 1}, the step was built for {self.num_class} classes")
            if loss.fused_filter:
                ctx.miner_seed = loss.hyp_miner.next_seed()
            ctx.T_grid = max(ctx.plan.T, T_cap)
        ctx.rot_seed = self._next_rot_seed() if own_rotation else None
        feat = self.nn_feat
        ctx.drop_seeds = None
        if own_dropout and getattr(feat, "dropout", 0.0) > 0.0 and feat.training and hasattr(feat, "next_drop_seed"):
            ctx.drop_seeds = (feat.next_drop_seed(), feat.next_drop_seed())
        ctx.radam_tables = self.optimizer.build_tables()
        blk.begin()
        blk.put_seeds(rot=ctx.rot_seed or 0, drop1=ctx.drop_seeds[0] if ctx.drop_seeds else 0,
                      drop2=ctx.drop_seeds[1] if ctx.drop_seeds else 0, miner=ctx.miner_seed or 0)
        if ctx.plan is not None:
            blk.put_plan(ctx.plan)
        blk.put_radam(ctx.radam_tables)
        blk.commit()
        ctx.seed_dev, ctx.toff_dev, ctx.kk_dev, ctx.radam_dev = blk.seed_dev, blk.toff_dev, blk.kk_dev, blk.radam_dev
        return ctx

    def forward_losses(self, points, label, targets, plan=None, rot=None, drop_masks=None, ctx=None):
        """points [B,N,3], label [B] category ids, targets [B,N] part ids (all on the device).  ctx: the step's StepCtx
        (training_step); without one the seeds are drawn here and passed by value."""
        if self.train_rotation == "so3":
            if ctx is not None and ctx.rot_seed is not None and rot is None:
                x = rotate_to_channels_first(points, seed=ctx.rot_seed, seed_dev=ctx.seed_dev["rot"])
            else:
                x = rotate_to_channels_first(points, rot=rot, seed=0 if rot is not None else self._next_rot_seed())
        else:
            x = points.transpose(2, 1).contiguous().float()
        decode = to_categorical(label, self.num_categories)
        if drop_masks is not None:
            x_euclidean = self.nn_feat(x, decode, drop_masks=drop_masks)
        elif ctx is not None and ctx.drop_seeds is not None:
            x_euclidean = self.nn_feat(x, decode, drop_seeds=((ctx.drop_seeds[0], ctx.seed_dev["drop1"]),
                                                              (ctx.drop_seeds[1], ctx.seed_dev["drop2"])))
        else:
            x_euclidean = self.nn_feat(x, decode)
        x_euclidean = x_euclidean.contiguous()      # ONE copy of the transposed head output: the embedder and the loss both want rows
        x_poincare = self.nn_emb(x_euclidean) if self.nn_emb is not None else x_euclidean
        xe = x_euclidean.view(-1, x_euclidean.shape[-1])
        xp = x_poincare.contiguous().view(-1, x_poincare.shape[-1])
        y = targets.reshape(-1).long()
        if ctx is not None and plan is None:
            plan = ctx.plan
        loss = self.metric_hyp_loss.compute_loss(xe, xp, y, plan=plan, step_ctx=ctx)
        out = {"loss_metric": loss["loss_metric"]["losses"],
               "loss_hyp": loss["loss_hyp"]["losses"] * self.trade_off}
        if self.metrics and hasattr(self.metric_hyp_loss, "loss_cosface"):
            # base_hyp_hc.py:126-130: accuracy / IoU of the CosFace logits every step (device scalars, no sync); the
            # logits are the ones the loss kernel just produced
            from .loss.ultrametric_loss import class_metrics
            acc, iou = class_metrics(self.metric_hyp_loss.loss_cosface.last_logits, y)
            out["acc"], out["iou"] = acc, iou
        return out

    # ---- test-time path (base_hyp_hc.py:81-86, 120-141, 194-200) ----
    def _decode_linkage(self, leaves_embeddings):
        """One cloud [N,E] -> linkage matrix as a host array, like the reference; batches: decode_linkage_batch."""
        return self.decode_linkage_batch(leaves_embeddings.unsqueeze(0))[0].cpu().numpy()

    def decode_linkage_batch(self, x_poincare):
        """x_poincare [B,N,E] -> Z [B,N-1,4] fp64 on the device (hpcs_complete_linkage: one workgroup per cloud)."""
        from .decode import decode_linkage
        return decode_linkage(x_poincare, self.metric_hyp_loss)

    @torch.no_grad()
    def test_step(self, points, label, targets, index="iou", rot=None):
        """forward(testing=True) + the scoring loop of test_step: losses, metrics, embeddings, linkage matrices and the
        per-cloud best cut, everything on the device; `score` is the batch mean of the best scores (base_hyp_hc.py:241)."""
        from .decode import optimal_k
        if self.test_rotation == "so3":                         # shapenet_hyp_hc.py:58-69
            x = rotate_to_channels_first(points, rot=rot, seed=0 if rot is not None else self._next_rot_seed())
        else:
            x = points.transpose(2, 1).contiguous().float()
        decode = to_categorical(label, self.num_categories)
        x_euclidean = self.nn_feat(x, decode)
        x_poincare = self.nn_emb(x_euclidean) if self.nn_emb is not None else x_euclidean
        y = targets.reshape(-1).long()
        loss = self.metric_hyp_loss.compute_loss(x_euclidean.reshape(-1, x_euclidean.shape[-1]),
                                                 x_poincare.reshape(-1, x_poincare.shape[-1]), y)
        out = {"loss_metric": loss["loss_metric"]["losses"], "loss_hyp": loss["loss_hyp"]["losses"] * self.trade_off}
        if hasattr(self.metric_hyp_loss, "loss_cosface"):
            from .loss.ultrametric_loss import class_metrics
            out["acc"], out["iou"] = class_metrics(self.metric_hyp_loss.loss_cosface.last_logits, y)
        Z = self.decode_linkage_batch(x_poincare)
        pred, best_k, best_score = optimal_k(targets, Z, index=index)
        out.update(x_euclidean=x_euclidean, x_poincare=x_poincare, linkage_matrix=Z, best_pred=pred, best_k=best_k,
                   best_score=best_score, score=best_score.float().mean())
        return out

    # ---- one optimisation step ------------------------------------------------------------------------------------
    def _backward_part(self, points, label, targets, ctx, rot, drop_masks):
        """zero grads -> forward -> losses -> backward (weight gradients on the side stream, joined at the end)."""
        zero_arena_reset(points.device, also=self.bucket)   # ONE zero launch: the fp64 accumulator arena + the gradient bucket
        self._prep_cache.begin_step()          # one launch re-lays out every layer's weights (recorded in the first step)
        try:
            if self.side_stream_wgrad and FORWARD_SIDE:          # the backward below is certain to follow and joins the stream
                with forward_side_stream():
                    losses = self.forward_losses(points, label, targets, rot=rot, drop_masks=drop_masks, ctx=ctx)
            else:
                losses = self.forward_losses(points, label, targets, rot=rot, drop_masks=drop_masks, ctx=ctx)
            total = losses["loss_metric"] + losses["loss_hyp"]
            if self.side_stream_wgrad:         # dW GEMMs on a side stream, joined before the all-reduce
                with deferred_weight_grads():
                    total.backward()
            else:
                total.backward()
        finally:
            self._prep_cache.end_step()        # the optimiser moves the weights next: nothing prepared stays valid
            zero_arena_release()
        losses["total_loss"] = total
        # detached: a loss tensor that outlives the step keeps its autograd graph -- and the AccumulateGrad nodes of every
        # parameter, bound to the stream they were created on -- alive, which breaks a later stream capture
        return {k_: v.detach() for k_, v in losses.items()}

    def _graph_eligible(self, targets_host, rot, drop_masks):
        loss = self.metric_hyp_loss
        if self.use_graph and self.graph_auto and self._graph is None and self._tune is None:
            # auto mode never captures under a multi-rank process group: stream capture beside RCCL's watchdog / proxy
            # threads has not run on hardware here (the world-2 test drives the two-graph path with gloo, HPCS_GRAPH=1),
            # and the multi-GPU step is GPU-bound, where eager measured faster anyway
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and \
                    __import__("os").environ.get("HPCS_GRAPH_MULTI", "0") != "1":
                # HPCS_GRAPH_MULTI=1: measure replay against eager under a multi-rank group too (two graphs around the eager
                # all-reduce; every step still issues exactly one collective, so ranks may even decide differently)
                self.use_graph = False
        return (self.use_graph and targets_host is not None and rot is None and drop_masks is None and loss.miner
                and loss.fused_filter and loss.hyp_miner.rng == "device" and loss.debug_triplets is None
                and self.train_rotation in ("so3", None) and self.training)

    def training_step(self, points, label, targets, targets_host=None, rot=None, drop_masks=None):
        """One optimisation step.  targets_host (CPU copy of the part labels) lets the miner's per-class plan be
        computed without a device->host sync in the middle of the step -- and makes the step replayable: after
        `graph_warmup` eager steps the steady-state step is captured ONCE as a hipGraph (forward, losses, backward,
        RAdam; with more than one rank: two graphs around the eager all-reduce) and every later step is: fill the
        parameter block (host, ~0.1 ms), one upload, one graph launch.  Two capture forms exist -- "side" (the weight
        gradients on their side stream: a fork / join graph) and "serial" (one stream: a linear graph, cheaper per node
        for small clouds) -- and `HPCS_GRAPH=auto` times both against the eager step.  The losses returned by a replayed
        step are the capture's output tensors, overwritten by every replay."""
        self._calls += 1
        if self._calls == 4:
            _gc_settle()
        if self._graph_eligible(targets_host, rot, drop_masks):
            if self._eager_steps >= self.graph_warmup:
                if not self.graph_auto:
                    return self._graphed_step(points, label, targets, targets_host)
                choice = self._auto_mode()
                if choice == "graph":
                    return self._timed(self._graphed_step, points, label, targets, targets_host)
                if choice == "eager_timed":
                    return self._timed(self._eager_step, points, label, targets, targets_host)
            else:
                self._eager_steps += 1
        return self._eager_step(points, label, targets, targets_host, rot, drop_masks)

    def _auto_mode(self):
        """auto mode schedule after the warm-up, per candidate: one lead-in step (for a capture form: the capture + its
        first replay) and then a BURST of 3 steps enqueued back to back with no host wait in between, timed as one event
        span from before the first to after the last -- what a training loop sees: the eager step's host work for step i+1
        overlaps the GPU's step i (timing single steps between host waits charged the eager step its enqueue latency and
        read both forms equal where the loop ran eager 3 % faster).  First the "side" form, then (that capture dropped: one
        private pool of activations at a time) the "serial" form, then eager; the fastest is kept for good (a winning "side"
        form is captured once more).  The serial form was withdrawn in round 2 because its replays corrupted the step's
        metric outputs; the cause was the runtime's handling of hipMemsetAsync NODES once a second graph exists (DESIGN.md
        "The stray store"), and the library issues none any more."""
        t = self._tune
        if t is None:
            t = self._tune = {"n": 0, "graph_side": [], "graph_serial": [], "eager": [], "pending": None, "choice": None}
        if t["choice"] is not None:
            return t["choice"]
        self._collect_timing()
        t["n"] += 1
        forms = self._candidate_forms()
        per = 4                                                           # lead-in + 3 timed
        if t["n"] <= per * len(forms):
            k, j = divmod(t["n"] - 1, per)
            if j == 0 and self._graph_form != forms[k]:
                self._drop_graph()
            self._graph_form = forms[k]
            t["kind"], t["pos"] = f"graph_{forms[k]}", (j, per)
            return "graph"
        if not self.use_graph:                                            # a capture failed: eager for good
            t["choice"] = "eager"
            return "eager"
        if t["n"] <= per * (len(forms) + 1):
            t["kind"], t["pos"] = "eager", (t["n"] - 1 - per * len(forms), per)
            return "eager_timed"
        ms = {f"graph_{f}": min(t[f"graph_{f}"]) for f in forms if t[f"graph_{f}"]}
        ms["eager"] = min(t["eager"])
        # a replayed step costs the host ~1.4 ms, the eager one 4-6 ms of a ~11 ms step: host jitter (a cold page cache, a busy
        # core) shows in the eager step only (r03: 10.9 ... 13.7 ms run to run on one box), so a capture within 2 % of the
        # eager step's best time is preferred
        best = min(ms, key=lambda k_: ms[k_] * (1.02 if k_ == "eager" else 1.0))
        t["measured_ms"] = ms
        if best == "eager":
            t["choice"] = "eager"
            self.use_graph = False
            self._drop_graph()               # its private pool holds a whole step's activations: released for the eager winner
        else:
            t["choice"] = "graph"
            self.graph_auto = False
            form = best[len("graph_"):]
            if form != self._graph_form:
                self._drop_graph()
                self._graph_form = form
        return t["choice"]

    def _candidate_forms(self):
        forced = __import__("os").environ.get("HPCS_GRAPH_FORM", "").lower()
        if forced in ("side", "serial"):
            return [forced]
        return ["side", "serial"] if self.side_stream_wgrad else ["serial"]

    def _drop_graph(self):
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph = None

    def graph_mode(self):
        """'graph' / 'eager' once decided (or forced), else None while the auto mode is still measuring."""
        if not self.use_graph:
            return "eager"
        if not self.graph_auto:
            return "graph"
        return None if self._tune is None or self._tune["choice"] is None else self._tune["choice"]

    def graph_variant(self):
        """'side' | 'serial': how the replayed step is captured (meaningful while graph_mode() == 'graph')."""
        return self._graph_form

    def _collect_timing(self):
        t = self._tune
        if t["pending"] is not None:
            kind, e0, e1, m = t["pending"]
            e1.synchronize()
            t[kind].append(e0.elapsed_time(e1) / m)
            t["pending"] = None

    def _timed(self, fn, *args):
        """Steps 1 .. per-1 of a candidate's schedule form one timed burst (step 0 is its lead-in)."""
        t = self._tune
        j, per = t["pos"]
        if j == 1:
            t["e0"] = torch.cuda.Event(enable_timing=True)
            t["e0"].record()
        out = fn(*args)
        if j == per - 1:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            t["pending"] = (t["kind"], t["e0"], e1, per - 1)
        return out

    # ---- host counters a step advances BEFORE its kernels run (restored when the step raises: ADVICE r2) -------------
    def _counters(self):
        loss, feat = self.metric_hyp_loss, self.nn_feat
        return (self.optimizer.param_groups[0].get("step", 0), self._rot_calls,
                loss.hyp_miner.seed if loss.miner else None, getattr(feat, "_drop_calls", None))

    def _restore_counters(self, c):
        loss, feat = self.metric_hyp_loss, self.nn_feat
        self.optimizer.param_groups[0]["step"], self._rot_calls = c[0], c[1]
        if c[2] is not None:
            loss.hyp_miner.seed = c[2]
        if c[3] is not None:
            feat._drop_calls = c[3]

    def _eager_step(self, points, label, targets, targets_host, rot=None, drop_masks=None):
        saved = self._counters()
        self._committed = False
        try:
            ctx = self._prepare(targets_host, own_rotation=(self.train_rotation == "so3" and rot is None),
                                own_dropout=drop_masks is None)
            return self._run_eager(points, label, targets, ctx, rot, drop_masks)
        except BaseException:
            self._rewind_or_poison(saved)
            raise

    def _rewind_or_poison(self, saved):
        """A step that raised BEFORE its optimiser launch left the device-side training state (weights, moments) untouched:
        RAdam's bias-correction count and the miner / rotation / dropout streams -- advanced on the host while the step was
        prepared -- are rewound, as the reference only advances them inside a step that ran (a caught OOM, a label check, a
        skipped batch).  Once the optimiser (or the replay that contains it) has been enqueued the device state HAS moved:
        rewinding the host counters then would desynchronise bias correction and the RNG streams from it, so
        they are kept and the step object is marked (`poisoned`: the failure is the caller's to handle)."""
        if getattr(self, "_committed", False):
            self.poisoned = True
        else:
            self._restore_counters(saved)

    def _run_eager(self, points, label, targets, ctx, rot=None, drop_masks=None):
        losses = self._backward_part(points, label, targets, ctx, rot, drop_masks)
        self.bucket.all_reduce_mean()
        self._committed = True                 # from here on the device-side weights / moments move
        self.optimizer.step(tables=ctx.radam_tables, tables_dev=ctx.radam_dev)
        _watch_disarm()
        return losses

    def _capture(self, points, label, targets, ctx, shapes):
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        g = {"shapes": shapes, "T_cap": int(ctx.plan.T * 1.25) + 1, "world": world, "form": self._graph_form,
             "in": (points.clone(), label.clone(), targets.clone())}
        ctx.T_grid = g["T_cap"]
        torch.cuda.synchronize()
        side = self.side_stream_wgrad
        self.side_stream_wgrad = side and self._graph_form == "side"
        try:
            g["fwd_bwd"], kept_raw = _new_graph()
            with torch.cuda.graph(g["fwd_bwd"], capture_error_mode="thread_local"):
                g["out"] = self._backward_part(*g["in"], ctx, None, None)
                if world == 1:
                    self.optimizer.step(tables=ctx.radam_tables, tables_dev=ctx.radam_dev)
            if world > 1:
                g["pool"] = g["fwd_bwd"].pool()
                g["opt"], _ = _new_graph()
                with torch.cuda.graph(g["opt"], pool=g["pool"], capture_error_mode="thread_local"):
                    self.optimizer.step(tables=ctx.radam_tables, tables_dev=ctx.radam_dev)
        finally:
            self.side_stream_wgrad = side
            _watch_disarm(freeze=True)
        # the source scan over csrc/ cannot see what torch's own ops put into the capture.  Walk the captured
        # graph(s): a memset NODE (hipMemsetAsync -- a torch fill of some shapes, the semaphores of a multi-block
        # reduction) loses its arguments once a second graph is instantiated on this runtime, so such a
        # capture is refused and the step stays eager.  memcpy nodes replay correctly (same reproducer) and are reported.
        g["nodes"] = {}
        for key in ("fwd_bwd", "opt"):
            if key in g:
                census = _lib_graph_census(g[key]) if kept_raw else None
                g["nodes"][key] = census
                if census is not None and census.get("memset", 0):
                    raise RuntimeError(f"the captured step holds {census['memset']} memset node(s) ({census}): refused")
                if kept_raw:
                    g[key].instantiate()           # keep_graph defers instantiation to here (after the inspection)
        self.last_capture_nodes = g["nodes"]
        return g

    def _graphed_step(self, points, label, targets, targets_host):
        saved = self._counters()
        self._committed = False
        try:
            g = self._graph
            shapes = (tuple(points.shape), tuple(label.shape), tuple(targets.shape), points.dtype, label.dtype, targets.dtype)
            T_cap = g["T_cap"] if g is not None else 0
            ctx = self._prepare(targets_host, own_rotation=self.train_rotation == "so3", own_dropout=True, T_cap=T_cap)
            if ctx.plan.T == 0:
                # nothing to sample (every class below two members): the loss takes its host-side path, not capturable
                return self._run_eager(points, label, targets, ctx)
            if g is not None and g["shapes"] != shapes:
                # another geometry (a last partial batch): eager, keeping the capture; only a geometry that stays is captured
                self._off_shape = self._off_shape + 1 if self._off_shape_key == shapes else 1
                self._off_shape_key = shapes
                if self._off_shape <= 3:
                    ctx.T_grid = ctx.plan.T
                    return self._run_eager(points, label, targets, ctx)
                g = None
            self._off_shape, self._off_shape_key = 0, None
            if g is not None and (ctx.plan.T > g["T_cap"] or g["form"] != self._graph_form):
                g = None                                         # more triplets than the grid covers / another form asked for
            if g is None:
                self._drop_graph()
                try:
                    g = self._graph = self._capture(points, label, targets, ctx, shapes)
                except Exception as ex:                          # noqa: BLE001 -- any capture failure ends in the eager step
                    import warnings
                    warnings.warn(f"hpcs_amd: hipGraph capture of the training step failed ({type(ex).__name__}: {ex}); "
                                  "continuing with the eager step")
                    self._graph = None
                    self.use_graph = False
                    torch.cuda.synchronize()
                    ctx.T_grid = ctx.plan.T
                    return self._run_eager(points, label, targets, ctx)
                # the capture itself executed nothing: fall through to the first replay with this step's parameters
            for dst, src in zip(g["in"], (points, label, targets)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            self._committed = True             # the replay contains the optimiser (world 1) / is followed by it
            g["fwd_bwd"].replay()
            if g["world"] > 1:
                self.bucket.all_reduce_mean()
                g["opt"].replay()
            self.metric_hyp_loss.last_T = ctx.plan.T
            return g["out"]
        except BaseException:
            self._rewind_or_poison(saved)
            raise
