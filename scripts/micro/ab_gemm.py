#!/usr/bin/env python3
"""gemm_x3_bench with an alternative build of the library: python scripts/micro/ab_gemm.py <lib.so>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hpcs_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
runpy.run_path(os.path.join(ROOT, "scripts", "gemm_x3_bench.py"), run_name="__main__")
