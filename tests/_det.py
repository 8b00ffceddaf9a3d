"""Counter-based deterministic test data (integer hash -> exactly representable fp32).

Shared by tests/golden/make_golden.py (which runs the reference on these inputs in
the build container) and by the parity tests (which regenerate the same inputs on
any machine), so fixtures only need to store the reference's OUTPUTS.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed):
    """fp32 tensor of the given shape, values k/2^23 - 1 in [-1, 1), reproducible bit for bit."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        h = _splitmix(_splitmix(ctr))
    v = (h >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def det_normalish(shape, seed):
    """Sum of four uniforms: bell-shaped, still exactly reproducible."""
    return (det_uniform(shape, seed) + det_uniform(shape, seed + 7919) + det_uniform(shape, seed + 104729)
            + det_uniform(shape, seed + 1299709)) * 0.5


def det_int(shape, seed, high):
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        h = _splitmix(_splitmix(ctr))
    return torch.from_numpy((h % np.uint64(high)).astype(np.int64).reshape(shape))


def det_cloud(B, N, seed, dup_every=0):
    """[B,3,N] unit-ball-normalised clouds (hpcs/utils/data.py:37-42 normalisation).

    dup_every>0 overwrites every dup_every-th point with its predecessor (exact duplicates,
    as sampling with replacement produces: hpcs/data/shapenet.py:109).
    """
    p = det_normalish((B, N, 3), seed).numpy().astype(np.float64)
    if dup_every:
        dst = np.arange(dup_every, N, dup_every)
        p[:, dst, :] = p[:, dst - 1, :]
    # float64 arithmetic then one rounding to fp32: insensitive to the summation order of the host
    p = p - p.mean(axis=1, keepdims=True)
    p = p / np.sqrt((p * p).sum(axis=2)).max(axis=1).reshape(B, 1, 1)
    return torch.from_numpy(np.ascontiguousarray(p.transpose(0, 2, 1)).astype(np.float32))


def det_params(shapes, seed):
    """Deterministic state dict for a {name: shape} map (see oracle.extractor.param_shapes)."""
    out = {}
    for i, (name, shp) in enumerate(sorted(shapes.items())):
        s = seed * 1000 + i
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_var"):
            out[name] = 1.0 + 0.25 * det_uniform(shp, s).abs()
        elif name.endswith("running_mean"):
            out[name] = 0.1 * det_uniform(shp, s)
        elif name.endswith("bn.weight") or (name.endswith(".weight") and len(shp) == 1):
            out[name] = 1.0 + 0.1 * det_uniform(shp, s)
        elif name.endswith(".bias"):
            out[name] = 0.1 * det_uniform(shp, s)
        else:
            fan_in = shp[1]
            out[name] = det_uniform(shp, s) * (1.0 / np.sqrt(fan_in)) * 1.5
    return out
