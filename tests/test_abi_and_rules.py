"""CPU-only: the C-ABI library loads and exports every declared symbol; the product never touches the oracle."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from hpcs_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.exported_symbols_declared_in_header()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert _lib.call("hpcs_version") >= 100


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before any HIP call (safe on a CPU-only box)."""
    from hpcs_amd import _lib
    L = _lib.lib()
    f = L.hpcs_knn
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert f(None, 1, 3, 8, 2, None, None, 0, None) == -4            # HPCS_ERR_NULL
    assert f(1, 1, 3, 8, 9, 1, None, 0, None) == -1                   # k > N: HPCS_ERR_SHAPE
    assert f(1, 1, 3, 128, 65, 1, None, 0, None) == -2                # k > 64: HPCS_ERR_UNSUPPORTED
    g = L.hpcs_gemm_nt_workspace
    g.restype = ctypes.c_size_t
    g.argtypes = [ctypes.c_int] * 5
    assert g(42, 42, 61440, 32, 1) > 0 and g(0, 1, 1, 1, 1) == 0


def test_product_fails_loudly_on_cpu_tensors():
    from hpcs_amd import _lib, ops
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from hpcs_amd.nn.hyperbolic import ExpMap
    with pytest.raises(_lib.HpcsNativeError):
        ops.knn(torch.zeros(1, 3, 16), 4)
    net = VN_DGCNN_partseg(3, 4, k=4, dropout=0.0, pooling="mean", num_categories=2)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 16), torch.zeros(1, 2))
    with pytest.raises(_lib.HpcsNativeError):
        ExpMap()(torch.zeros(2, 4))


def test_state_dict_keys_match_reference_layout():
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    from oracle.extractor import param_shapes
    for pooling in ("mean", "max"):
        net = VN_DGCNN_partseg(3, 32, k=20, dropout=0.5, pooling=pooling, num_categories=16)
        sd = net.state_dict()
        want = param_shapes(32, 16, pooling)
        assert set(sd) == set(want)
        for k_, shp in want.items():
            assert tuple(sd[k_].shape) == tuple(shp), k_
    assert sum(p.numel() for p in net.parameters()) > 1_300_000


def _py_files(d):
    for base, _, files in os.walk(d):
        if "__pycache__" in base:
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_oracle_is_test_infrastructure_only():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|from\s+\.+\s*oracle\b", re.M)
    for f in _py_files(os.path.join(ROOT, "hpcs_amd")):
        assert not pat.search(open(f).read()), f"{f} imports the oracle"
    # bench.py may use it only inside cpu_baseline(); __graft_entry__ only inside build()/smoke()
    src = open(os.path.join(ROOT, "bench.py")).read()
    for m in pat.finditer(src):
        assert "def cpu_baseline" in src[:m.start()] and "def main" not in src[src.index("def cpu_baseline"):m.start()]
    assert "/root/reference" not in src
    for f in list(_py_files(os.path.join(ROOT, "hpcs_amd"))) + [os.path.join(ROOT, "__graft_entry__.py")]:
        assert "/root/reference" not in open(f).read(), f
    assert "TEST INFRASTRUCTURE" in open(os.path.join(ROOT, "oracle", "__init__.py")).read()


def _compatible(ref_params, got):
    """Every call that binds under the reference's signature binds under ours with the same meaning: the reference's
    named parameters appear in the same order with the same defaults; whatever we add has a default (or is **kwargs).
    A reference `**kwargs` (PML base-class arguments) needs `**kwargs` or extra defaulted keywords on our side; a
    reference `*args` (forwarded to a PML constructor) may be spelled out as named parameters."""
    import inspect
    ref_named = [(n, d) for n, kind, d in ref_params if not kind.startswith("VAR_")]
    ref_var_kw = any(kind == "VAR_KEYWORD" for _, kind, _ in ref_params)
    ref_var_pos = any(kind == "VAR_POSITIONAL" for _, kind, _ in ref_params)
    gp = list(got.parameters.values())
    g_named = [p_ for p_ in gp if p_.kind in (p_.POSITIONAL_OR_KEYWORD, p_.KEYWORD_ONLY, p_.POSITIONAL_ONLY)]
    g_var_kw = any(p_.kind == p_.VAR_KEYWORD for p_ in gp)
    names = [p_.name for p_ in g_named]
    pos = -1
    for n, d in ref_named:
        if n not in names:
            return f"parameter {n!r} missing"
        i = names.index(n)
        if i < pos:
            return f"parameter {n!r} out of order"
        pos = i
        gd = g_named[i].default
        if d is not None and (gd is inspect.Parameter.empty or repr(gd) != d):
            return f"default of {n!r}: reference {d}, ours {gd!r}"
        # (a default ADDED where the reference has none keeps every reference call valid)
    ref_names = {n for n, _ in ref_named}
    for p_ in g_named:
        if p_.name not in ref_names and p_.default is inspect.Parameter.empty and not ref_var_pos:
            return f"extra required parameter {p_.name!r}"
    if ref_var_kw and not g_var_kw and not any(p_.name not in ref_names for p_ in g_named):
        return "reference takes **kwargs, ours takes no further keywords"
    return None


def test_operator_api_signatures_match_reference():
    """tests/golden/api_signatures.json = the parameter lists (name, kind, default) of the reference's importable
    hot-path symbols (generated by tests/golden/make_golden_r2.py, stored as data); every one must exist in hpcs_amd
    (hpcs.X -> hpcs_amd.X) and accept every call the reference's signature accepts (SURVEY 8b drop-in boundary)."""
    import importlib
    import inspect
    import json
    sig = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    assert len(sig) >= 60
    problems = []
    for key, ref_params in sorted(sig.items()):
        parts = key.split(".")
        mod, rest = None, None
        for i in range(len(parts), 0, -1):
            try:
                mod = importlib.import_module(".".join(parts[:i]).replace("hpcs", "hpcs_amd", 1))
                rest = parts[i:]
                break
            except ImportError:
                continue
        obj = mod
        try:
            for r in rest:
                obj = obj.__dict__[r] if (inspect.isclass(obj) and r in obj.__dict__) else getattr(obj, r)
            got = inspect.signature(obj)
        except (AttributeError, KeyError, TypeError, ValueError) as e:
            problems.append(f"{key}: missing ({e!r})")
            continue
        why = _compatible(ref_params, got)
        if why:
            problems.append(f"{key}: {why}\n      reference {ref_params}\n      ours      {got}")
    assert not problems, "\n".join(problems)


def test_library_never_issues_memset_or_memcpy_nodes():
    """Round 3 (DESIGN.md "The stray store"): memset NODES of a replayed hipGraph lose their arguments on this runtime once a
    second graph exists (stale fill pattern / stale destination -> corrupted outputs, then a GPU memory-access fault).  The
    library therefore zero-fills with its own kernel (csrc/common.h hpcs_zero_async) and must not call the runtime's
    memset / memcpy entry points on the launch path."""
    import re
    csrc = os.path.join(ROOT, "hpcs_amd", "csrc")
    # every hipMemset* / hipMemcpy* spelling (2D / 3D / D8 / D32 / peer ...); hipMemcpyToSymbol is a host-side, stream-less
    # store of the profiling stamps' address, never captured
    pat = re.compile(r"\bhipMem(set|cpy)(?!ToSymbol)\w*\s*\(")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, f)), 1):
            code = line.split("//")[0]
            if pat.search(code):
                hits.append(f"{f}:{ln}: {line.strip()}")
    assert hits == [], hits
