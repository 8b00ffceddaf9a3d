"""Generates tests/golden/hyperbolic_lca.npz by running the REFERENCE's HyperbolicLCA (hpcs/distances/poincare.py:190-237)
on seeded inputs.  Same import mechanism and rules as make_golden.py (the reference is imported and called, never copied;
PML supplies only the base-class NAME BaseDistance through a placeholder).  No-op when /root/reference is absent."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def inputs():
    from _det import det_normalish
    x = det_normalish((37, 2), 8101) * 0.45
    y = det_normalish((23, 2), 8102) * 0.45
    x[0] = 0.0                       # the origin
    x[1] = torch.tensor([3.0, -2.0])  # outside the ball: project() clips to radius 1 - 4e-3
    y[0] = x[2]                      # coincident points
    y[1] = torch.tensor([0.9, 0.4])
    y[2] = -x[3]
    return x, y


def main():
    import make_golden as MG
    if not os.path.isdir(MG.REF):
        print("reference absent: nothing to do")
        return
    MG._import_reference()
    from hpcs.distances import poincare as DP
    torch.set_num_threads(2)
    x, y = inputs()
    lca = DP.HyperbolicLCA()
    out = {"x": x.numpy(), "y": y.numpy(),
           "mat_xy": lca.compute_mat(x, y).numpy(),
           "mat_xx": lca.compute_mat(x, None).numpy(),
           # pairwise_distance uses Python's builtin min() on tensors (poincare.py:236): it only runs for ONE pair at a time
           "pairs": torch.cat([lca.pairwise_distance(x[i:i + 1], y[i:i + 1]).reshape(-1) for i in range(23)]).numpy()}
    np.savez_compressed(os.path.join(HERE, "hyperbolic_lca.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
