"""Builds the C part of the oracle (TEST INFRASTRUCTURE): oracle/libhpcs_oracle.so."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    src = os.path.join(_HERE, "knn_oracle.c")
    out = os.path.join(_HERE, "libhpcs_oracle.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", out, src, "-lm"]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True))
