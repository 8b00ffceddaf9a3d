"""ctypes binding of libhpcs_mi355x.so (include/hpcs_mi355x.h).

There is NO CPU fallback: if the shared library is missing, or a tensor is not on a
HIP device, the call raises.  The oracle (oracle/) is never imported from here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhpcs_mi355x.so")

_P, _I, _L, _F, _Z, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint64
_KIND = {"p": _P, "i": _I, "l": _L, "f": _F, "z": _Z, "u": _U, "s": _P}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "hpcs_mi355x.h")


def _parse_header():
    """Derives every entry point's ctypes signature from include/hpcs_mi355x.h (single source of truth).

    kinds: p pointer, i int/int32, l int64, u uint64, z size_t, f float, s stream (filled in automatically).
    """
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for ret, name, params in re.findall(r"\b(int|size_t)\s+(hpcs_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        kinds = ""
        for prm in [q.strip() for q in params.split(",")]:
            if prm in ("", "void"):
                continue
            if "hpcs_stream_t" in prm:
                kinds += "s"
            elif "*" in prm:
                kinds += "p"
            elif "uint64_t" in prm:
                kinds += "u"
            elif "int64_t" in prm:
                kinds += "l"
            elif "size_t" in prm:
                kinds += "z"
            elif "float" in prm:
                kinds += "f"
            elif re.search(r"\b(int|int32_t)\b", prm):
                kinds += "i"
            else:
                raise RuntimeError(f"cannot map parameter '{prm}' of {name}")
        sigs[name] = (kinds, "z" if ret == "size_t" else "i")
    return sigs


SIGNATURES = _parse_header()

_lib = None


class HpcsNativeError(RuntimeError):
    pass


def lib():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HpcsNativeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hpcs_amd has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


_bound = {}


def _fn(name):
    f = _bound.get(name)
    if f is None:
        args, ret = SIGNATURES[name]
        f = getattr(lib(), name)       # AttributeError here = header/library mismatch: fail loudly
        f.argtypes = [_KIND[c] for c in args]
        f.restype = _KIND[ret]
        _bound[name] = f
    return f


_STATUS = {-1: "bad shape", -2: "unsupported configuration", -3: "workspace too small", -4: "null pointer"}


def _dev_ptr(t):
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TypeError(f"expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise HpcsNativeError("hpcs_amd kernels need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise HpcsNativeError("hpcs_amd kernels need contiguous tensors")
    return t.data_ptr()


# Optional per-entry-point timing with events on the launch stream (bench.py's roofline leg).
# PROFILE = {"hpcs_knn": [(start_event, end_event, tag), ...], ...}; only names present as keys are timed.
PROFILE = None
PROFILE_FILTER = None      # optional callable(name, shape_tag) -> bool: time only the launches it accepts
EVENT_POOL = None          # optional list of pre-created timing events (popped; creating them inside a timed loop perturbs it)


class Watch:
    """Diagnostics: a stream-ordered watchpoint (csrc/debug.hip).  Once armed on a device region, one compare kernel is
    enqueued after EVERY later entry-point call (also inside a hipGraph capture, also from autograd's backward thread);
    status[0] then names the first call after which the region no longer matched its snapshot: `names[status[0] - 1]`.
    Nothing is allocated after construction, so the memory layout of the run under test does not change."""

    def __init__(self, device, nbytes=512):
        self.nbytes = int(nbytes)
        self.shadow = torch.zeros(self.nbytes // 4, dtype=torch.int32, device=device)
        self.status = torch.zeros(8, dtype=torch.int32, device=device)
        self.names = []
        self.ptr = None
        self.armed = False
        self.frozen = False        # True: keep the names of the first armed pass (a capture) and stop recording

    def arm(self, ptr):
        """Snapshot the region at `ptr` (device address) and start watching it."""
        self.ptr = int(ptr)
        _call("hpcs_debug_snapshot", self.ptr, self.shadow, self.nbytes)
        if not self.frozen:
            self.names = []
        self._seq = 0
        self.armed = True

    def disarm(self):
        self.armed = False

    def check(self, label):
        """One compare kernel now (torch-op boundaries that no entry point brackets)."""
        if not self.armed:
            return
        self._seq += 1
        if not self.frozen:
            self.names.append(label)
        _call("hpcs_debug_watch", self.ptr, self.shadow, self.nbytes, self._seq, self.status)

    def report(self):
        st = [int(v) for v in self.status.cpu()]
        if st[0] == 0:
            return None
        seq = st[0]
        return {"seq": seq, "after": self.names[seq - 1] if 0 < seq <= len(self.names) else None,
                "previous": self.names[seq - 2] if 1 < seq <= len(self.names) + 1 else None,
                "word": st[1], "found": st[2] & 0xffffffff, "expected": st[3] & 0xffffffff, "differing_words": st[4],
                "next_word": st[5] & 0xffffffff, "checks": st[6], "check_at_hit": st[7], "calls_per_pass": len(self.names)}


WATCH = None               # a Watch instance while a hunt is on (scripts/fault_hunt.py); None in every normal run


def call(name, *args, tag=None):
    """Calls an entry point; tensors become device pointers, the stream argument is appended."""
    if WATCH is not None and WATCH.armed and not name.endswith("_workspace"):
        rc = _call(name, *args)
        WATCH.check((name, tag if tag is not None else _shape_tag(name, args)))
        return rc
    if PROFILE is not None and name in PROFILE:
        tg = tag if tag is not None else _shape_tag(name, args)
        if PROFILE_FILTER is None or PROFILE_FILTER(name, tg):
            if EVENT_POOL is not None and len(EVENT_POOL) >= 2:
                e0, e1 = EVENT_POOL.pop(), EVENT_POOL.pop()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = _call(name, *args)
            e1.record()
            PROFILE[name].append((e0, e1, tg))
            return rc
    return _call(name, *args)


def _shape_tag(name, args):
    """The integer (non-pointer) arguments of a call, in order (pointers may arrive as raw ints: skipped by kind)."""
    kinds = [c for c in SIGNATURES[name][0] if c != "s"]
    return tuple(a for c, a in zip(kinds, args) if c != "p" and isinstance(a, int) and not isinstance(a, bool))


def _raw_stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object on every call (~9 us: with ~320 launches per step it was a tenth of the step's host time); the raw accessor
    is the one torch's own extension launchers use."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                     # older / differently built torch: the public, slower route
        return torch.cuda.current_stream().cuda_stream


_RAW_INT = frozenset(("hpcs_version", "hpcs_vn_rc_supported", "hpcs_debug_graph_census"))     # int results, not statuses


def _call(name, *args):
    kinds, ret = SIGNATURES[name]
    fn = _fn(name)
    conv = []
    ai = 0
    for c in kinds:
        if c == "s":
            conv.append(_raw_stream())
            continue
        a = args[ai]
        ai += 1
        if c == "p":
            conv.append(a if (a is None or isinstance(a, int)) else _dev_ptr(a))
        elif c == "f":
            conv.append(float(a))
        else:
            conv.append(int(a))
    if ai != len(args):
        raise TypeError(f"{name}: expected {ai} arguments, got {len(args)}")
    rc = fn(*conv)
    if ret == "i" and name not in _RAW_INT and rc != 0:
        msg = _STATUS.get(rc, f"HIP error {-(rc + 1000)}" if rc <= -1000 else "unknown")
        raise HpcsNativeError(f"{name} failed: {msg} (status {rc})")
    return rc


GRAPH_NODE_TYPES = ("kernel", "memcpy", "memset", "host", "child_graph", "empty", "wait_event", "event_record",
                    "ext_sem_signal", "ext_sem_wait", "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol",
                    "batch_mem_op", "other")


def graph_census(cuda_graph):
    """{node type: count} of a captured torch.cuda.CUDAGraph built with keep_graph=True (hpcs_debug_graph_census walks
    hipGraphGetNodes / hipGraphNodeGetType, child graphs included).  None when this torch cannot hand out the raw graph."""
    raw = getattr(cuda_graph, "raw_cuda_graph", None)
    if raw is None:
        return None
    try:
        ptr = int(raw())
    except Exception:                                   # noqa: BLE001 -- built without keep_graph / older torch
        return None
    counts = (ctypes.c_int64 * 16)()
    rc = _fn("hpcs_debug_graph_census")(ctypes.c_void_p(ptr), ctypes.cast(counts, ctypes.c_void_p))
    if rc < 0:
        raise HpcsNativeError(f"hpcs_debug_graph_census failed (status {rc})")
    out = {n: int(c) for n, c in zip(GRAPH_NODE_TYPES, counts) if c}
    out["total"] = int(rc)
    return out


KNN_KERNEL_CODES = {"auto": 0, "tiled": 1, "queue": 2, "bq": 4, "bq2": 5}


def set_tuning(name, value):
    """Tests / A-B scripts: replace one value of the library's tuning table (read from the HPCS_* environment once at first
    use, immutable otherwise).  name = the environment variable's name, value = its integer code."""
    fn = _fn("hpcs_debug_set_tuning")
    rc = fn(name.encode(), int(value))
    if rc != 0:
        raise HpcsNativeError(f"hpcs_debug_set_tuning({name!r}) failed (status {rc})")


def exported_symbols_declared_in_header():
    """Names declared in include/hpcs_mi355x.h (used by the ABI test)."""
    return sorted(SIGNATURES)
