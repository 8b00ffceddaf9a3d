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
from .ops import deferred_weight_grads, zero_arena_release, zero_arena_reset
from .optim import RAdam


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


def rotate_to_channels_first(points, rot=None, seed=0, return_rot=False):
    """points [B,N,3] -> [B,3,N] = transpose(points @ R) with one Haar rotation per cloud, in ONE launch
    (shapenet_hyp_hc.py:63-73).  rot [B,3,3] given: use it; else quaternions from the counter RNG keyed by `seed`."""
    from . import _lib
    if not points.is_cuda:
        raise _lib.HpcsNativeError("hpcs_amd needs HIP-device tensors (no CPU fallback)")
    pts = points.contiguous() if points.dtype == torch.float32 else points.float().contiguous()
    B, N, _ = pts.shape
    x = torch.empty(B, 3, N, device=pts.device, dtype=torch.float32)
    R = torch.empty(B, 3, 3, device=pts.device, dtype=torch.float32) if return_rot else None
    rin = None if rot is None else rot.to(torch.float32).contiguous()
    _lib.call("hpcs_so3_rotate_points", pts, B, N, int(seed) & (2 ** 63 - 1), rin, x, R)
    return (x, R) if return_rot else x


def to_categorical(y, num_classes):
    """1-hot encode (hpcs/utils/data.py:24-29), on the device of y."""
    return torch.nn.functional.one_hot(y.long(), num_classes).float()


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
        self._rot_seed = int(torch.initial_seed()) & (2 ** 62 - 1)
        self._rot_calls = 0
        self.side_stream_wgrad = __import__("os").environ.get("HPCS_SIDE_STREAM", "1") != "0"   # 0: serial (profiling)
        self.metrics = True
        self.optimizer = None
        self.bucket = None

    def configure(self):
        """Call once the module is on its device: flat gradient bucket + RAdam over ALL parameters (base_hyp_hc.py:143)."""
        params = list(self.parameters())
        self.bucket = FlatGradBucket(params)
        self.optimizer = RAdam(params, lr=self.lr)
        return self

    def forward_losses(self, points, label, targets, plan=None, rot=None, drop_masks=None):
        """points [B,N,3], label [B] category ids, targets [B,N] part ids (all on the device)."""
        if self.train_rotation == "so3":
            self._rot_calls += 1            # counter RNG stream: (torch seed at configure time, call index)
            x = rotate_to_channels_first(points, rot=rot, seed=self._rot_seed + self._rot_calls)
        else:
            x = points.transpose(2, 1).contiguous().float()
        decode = to_categorical(label, self.num_categories)
        x_euclidean = self.nn_feat(x, decode) if drop_masks is None else self.nn_feat(x, decode, drop_masks=drop_masks)
        x_poincare = self.nn_emb(x_euclidean) if self.nn_emb is not None else x_euclidean
        xe = x_euclidean.contiguous().view(-1, x_euclidean.shape[-1])
        xp = x_poincare.contiguous().view(-1, x_poincare.shape[-1])
        y = targets.reshape(-1).long()
        loss = self.metric_hyp_loss.compute_loss(xe, xp, y, plan=plan)
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
            self._rot_calls += 1
            x = rotate_to_channels_first(points, rot=rot, seed=self._rot_seed + self._rot_calls)
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

    def training_step(self, points, label, targets, targets_host=None, rot=None, drop_masks=None):
        """One optimisation step.  targets_host (CPU copy of the part labels) lets the miner's per-class plan be
        computed without a device->host sync in the middle of the step."""
        plan = None
        if targets_host is not None and self.metric_hyp_loss.miner:
            plan = self.metric_hyp_loss.make_plan(targets_host)
        self.bucket.zero()
        zero_arena_reset(points.device)        # one memset for all the step's fp64 accumulators
        losses = self.forward_losses(points, label, targets, plan=plan, rot=rot, drop_masks=drop_masks)
        total = losses["loss_metric"] + losses["loss_hyp"]
        if self.side_stream_wgrad:             # dW GEMMs on a side stream, joined before the all-reduce
            with deferred_weight_grads():
                total.backward()
        else:
            total.backward()
        zero_arena_release()
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        losses["total_loss"] = total.detach()
        return losses
