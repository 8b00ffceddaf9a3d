"""`hpcs_amd.install_as("hpcs")`: make the reference's import lines resolve to the MI355X mirror, unchanged.

    import hpcs_amd; hpcs_amd.install_as("hpcs")
    from hpcs.nn.dgcnn import VN_DGCNN_partseg                    # train.py:12-20
    from hpcs.nn.hyperbolic import ExpMap, MLPExpMap
    from hpcs.optim import RAdam                                   # hpcs/models/base_hyp_hc.py:16-21
    from hpcs.distances.poincare import project
    from hpcs.loss.ultrametric_loss import MetricHyperbolicLoss

How: every hot-path module of the mirror is registered in `sys.modules` under the reference's module path.  Leaf modules
are the mirror's module OBJECTS (one copy of every class and of every module-level switch); packages are thin proxies that
re-export the mirror package's names and keep the reference's directory on their `__path__` when the reference package is
importable, so everything OUTSIDE the hot path (`hpcs.data`, `hpcs.models`, `hpcs.utils.viz`, `hpcs.nn.pointnet`, the
non-VN `DGCNN_partseg` of `hpcs.nn.dgcnn` ...) still loads from the reference's own files -- SURVEY.md section 2 marks
those out of scope, they are not re-implemented here.  A name a proxy does not have is looked up in the reference's
sibling modules of that package (`from hpcs.nn.dgcnn import DGCNN_partseg`).  Without the reference on `sys.path` the
out-of-scope names raise ImportError / AttributeError naming this fact.
"""
import importlib
import importlib.util
import os
import sys
import types

# reference module path (relative to the package) -> True for packages
_MIRRORED = {
    "nn": True, "nn.dgcnn": True, "nn.dgcnn.vn_dgcnn_partseg": False, "nn.dgcnn.utils": True,
    "nn.dgcnn.utils.vn_layers": False, "nn.dgcnn.utils.vn_dgcnn_util": False,
    "nn.hyperbolic": True, "nn.hyperbolic.hyp_embed": False,
    "distances": True, "distances.cosine": False, "distances.lca": False, "distances.poincare": False,
    "miner": True, "miner.loss_and_miner_utils": False, "miner.triplet_margin_miner": False,
    "miner.triplet_margin_loss": False,
    "loss": True, "loss.ultrametric_loss": False, "loss.hierarchical_cosface_loss": False,
    "optim": True, "optim.radam": False,
    "utils": True, "utils.poincare": False, "utils.math": False,
}
# leaf modules of which the mirror implements only the hot-path part: proxies with a fallback to the reference's file
_PARTIAL_LEAVES = {"utils.scores", "utils.data"}

_installed = {}


def _reference_root(name, reference_root):
    """Directory of the reference's `<name>` package, or None.  Only a real on-disk package that is not this mirror."""
    if reference_root is not None:
        cand = os.path.join(reference_root, name)
        return cand if os.path.isdir(cand) else None
    for p in sys.path:
        cand = os.path.join(p or ".", name)
        if os.path.isfile(os.path.join(cand, "__init__.py")) and os.path.isdir(os.path.join(cand, "nn", "dgcnn")):
            return os.path.abspath(cand)
    return None


class _Proxy(types.ModuleType):
    """Package (or partial leaf) of the overlay: the mirror's names first, then the reference's sibling modules."""

    def __getattr__(self, attr):                                   # only reached when normal lookup failed
        if attr.startswith("__"):
            raise AttributeError(attr)
        errors = []
        for cand in self.__dict__.get("_hpcs_fallback", ()):       # module names or file paths in the reference
            try:
                mod = cand() if callable(cand) else importlib.import_module(cand)
            except Exception as ex:                                # noqa: BLE001 -- a missing third-party dependency
                errors.append(f"{getattr(cand, '__name__', cand)}: {type(ex).__name__}: {ex}")
                continue
            if hasattr(mod, attr):
                value = getattr(mod, attr)
                setattr(self, attr, value)
                return value
        why = ("; tried " + "; ".join(errors)) if errors else ""
        raise AttributeError(f"{self.__name__}.{attr} is outside the MI355X hot path (SURVEY.md section 2) and the reference "
                             f"package does not provide it here{why}")


def install_as(name="hpcs", reference_root=None):
    """Registers the mirror under the reference's module paths (idempotent).  reference_root: directory that CONTAINS
    the reference's `hpcs/` package (default: found on sys.path, else none -- then only the hot path resolves).
    Returns the list of module names registered."""
    if name in _installed:
        return list(_installed[name])
    if name in sys.modules and not isinstance(sys.modules[name], _Proxy):
        # the real package was imported first: its hot-path submodules may already be loaded -- overlay them anyway
        pass
    ref = _reference_root(name, reference_root)
    if reference_root is not None and ref is not None and reference_root not in sys.path:
        sys.path.append(reference_root)
    names = []
    top = sys.modules.get(name)
    if top is None:
        if ref is not None:
            spec = importlib.util.spec_from_file_location(name, os.path.join(ref, "__init__.py"), submodule_search_locations=[ref])
            top = importlib.util.module_from_spec(spec)
            sys.modules[name] = top
            spec.loader.exec_module(top)
        else:
            top = _Proxy(name)
            top.__path__ = []
            top.__package__ = name
            sys.modules[name] = top
    names.append(name)

    def ref_dir(rel):
        d = os.path.join(ref, *rel.split(".")) if ref else None
        return d if d and os.path.isdir(d) else None

    def ref_file(rel):
        f = os.path.join(ref, *rel.split(".")) + ".py" if ref else None
        return f if f and os.path.isfile(f) else None

    def loader_of_file(path, private_name):
        def load():
            if private_name in sys.modules:
                return sys.modules[private_name]
            spec = importlib.util.spec_from_file_location(private_name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[private_name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                del sys.modules[private_name]
                raise
            return mod
        load.__name__ = path
        return load

    def attach(full, mod):
        sys.modules[full] = mod
        parent, _, child = full.rpartition(".")
        setattr(sys.modules[parent], child, mod)
        names.append(full)

    for rel in sorted(set(_MIRRORED) | _PARTIAL_LEAVES, key=lambda r: (r.count("."), r)):
        full = f"{name}.{rel}"
        mirror = importlib.import_module(f"hpcs_amd.{rel}")
        is_pkg = _MIRRORED.get(rel, False)
        if not is_pkg and rel not in _PARTIAL_LEAVES:
            attach(full, mirror)                                   # the mirror's own module object
            continue
        proxy = _Proxy(full, mirror.__doc__)
        proxy.__package__ = full if is_pkg else full.rpartition(".")[0]
        for k, v in vars(mirror).items():
            if not k.startswith("__"):
                setattr(proxy, k, v)
        if hasattr(mirror, "__all__"):
            proxy.__all__ = list(mirror.__all__)
        fallback = []
        if is_pkg:
            d = ref_dir(rel)
            proxy.__path__ = [d] if d else []
            if d:                                                  # sibling modules of the reference package, not mirrored
                for f in sorted(os.listdir(d)):
                    sub = f[:-3] if f.endswith(".py") else (f if os.path.isdir(os.path.join(d, f)) and
                                                            os.path.isfile(os.path.join(d, f, "__init__.py")) else None)
                    if sub and sub != "__init__" and f"{rel}.{sub}" not in _MIRRORED:
                        fallback.append(f"{full}.{sub}")
        else:
            f = ref_file(rel)
            if f:
                fallback.append(loader_of_file(f, f"_hpcs_reference_.{rel}"))
        proxy.__dict__["_hpcs_fallback"] = fallback
        attach(full, proxy)
    _installed[name] = names
    return list(names)


def uninstall(name="hpcs"):
    """Removes everything install_as(name) registered (tests)."""
    for full in _installed.pop(name, ()):
        sys.modules.pop(full, None)
    for k in [k for k in sys.modules if k.startswith("_hpcs_reference_.") or k == name or k.startswith(name + ".")]:
        sys.modules.pop(k, None)
