"""hpcs_amd -- MI355X-native implementation of the HPCS training hot path.

Mirrors the reference's operator API (hpcs.nn / hpcs.distances / hpcs.miner / hpcs.loss /
hpcs.optim) on top of hand-written gfx950 kernels in csrc/ (C ABI: include/hpcs_mi355x.h).
"""
__version__ = "0.1.0"
