"""Generates tests/golden/decode.npz: the test-time decode of row f4 computed by the REFERENCE side.

  * linkage matrices: the three statements of BaseSimilarityHypHC._decode_linkage (hpcs/models/base_hyp_hc.py:81-86) --
    normalize_embeddings (F.normalize * clamp(scale), ultrametric_loss.py:139-143), the reference's own `project`
    (imported from hpcs.distances.poincare) and scipy.cluster.hierarchy.linkage(method='complete', metric='cosine').
    The method itself sits on a LightningModule (pytorch_lightning is absent), so its three calls are made here;
  * get_optimal_k: the reference function, imported from /root/reference/hpcs/utils/scores.py and called as test_step
    does ('iou') and with its other index ('ri').
Inputs come from tests/_det.py; only outputs are stored.  No-op when /root/reference is absent."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

CASES = {            # name: (N, E, seed, scale, number of label values, duplicate stride)
    "a": (256, 32, 9101, 1e-3, 4, 0),
    "b": (200, 2, 9102, 0.5, 3, 5),          # exact duplicate / opposite directions: ties at height 0 and 2
    "c": (128, 7, 9103, 2.0, 6, 0),          # odd E; scale clamps to 1 and project() clips the radius
}


def inputs(name):
    from _det import det_int, det_normalish
    N, E, seed, scale, nlab, dup = CASES[name]
    x = det_normalish((N, E), seed)
    if dup:
        x[::dup] = x[0]
        x[1::dup] = -x[0]
    y = det_int((N,), seed + 1, nlab) * 3 + 1            # label values with gaps: remap_labels matters
    # spatially coherent labels make the scores non-trivial: label by the sign pattern of the first coordinates
    y = ((x[:, 0] > 0).long() + 2 * (x[:, 1 % E] > 0).long()) % nlab * 3 + 1 if name != "c" else y
    return x, y, scale


def main():
    import make_golden as MG
    if not os.path.isdir(MG.REF):
        print("reference absent: nothing to do")
        return
    MG._import_reference()
    from hpcs.distances.poincare import project
    from scipy.cluster.hierarchy import linkage
    spec = importlib.util.spec_from_file_location("ref_scores", os.path.join(MG.REF, "hpcs/utils/scores.py"))
    scores = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scores)
    torch.set_num_threads(2)
    out = {}
    for name in CASES:
        x, y, scale = inputs(name)
        s = torch.tensor([scale])
        e = torch.nn.functional.normalize(x, p=2, dim=1) * s.clamp_min(1e-4).clamp_max(1.0)
        leaves = project(e).detach().cpu()
        Z = linkage(leaves, method="complete", metric="cosine")
        out[f"{name}_leaves"] = leaves.numpy()
        out[f"{name}_Z"] = Z
        for index in ("iou", "ri"):
            pred, k, score = scores.get_optimal_k(y.clone(), Z, index)
            out[f"{name}_{index}_pred"] = np.asarray(pred).astype(np.int32)
            out[f"{name}_{index}_k"] = np.int64(k)
            out[f"{name}_{index}_score"] = np.float64(score)
    path = os.path.join(HERE, "decode.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if v.ndim})


if __name__ == "__main__":
    main()
