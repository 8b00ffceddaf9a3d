"""CPU-only: the host side of the C-ABI library (argument validation, workspace / ticket sizing, launch plans) built with
AddressSanitizer and driven without a GPU (scripts/asan_host_check.py; SURVEY section 5 -- sanitizers on the CPU build
only, GPU ASan is not available on this pool).  ~1 minute of hipcc."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_launch_shims_are_clean_under_address_sanitizer():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asan_host_check.py")], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and "no report" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
