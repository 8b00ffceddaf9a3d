"""hpcs_amd -- MI355X-native implementation of the HPCS training hot path.

Mirrors the reference's operator API (hpcs.nn / hpcs.distances / hpcs.miner / hpcs.loss /
hpcs.optim) on top of hand-written gfx950 kernels in csrc/ (C ABI: include/hpcs_mi355x.h).
"""
__version__ = "0.1.0"


def install_as(name="hpcs", reference_root=None):
    """Registers the mirror's modules under the reference's module paths, so `from hpcs.loss.ultrametric_loss import
    MetricHyperbolicLoss` etc. (train.py:12-20, hpcs/models/base_hyp_hc.py:16-21) resolve unchanged (hpcs_amd/_alias.py)."""
    from ._alias import install_as as _install
    return _install(name, reference_root)
