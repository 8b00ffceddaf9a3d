"""torch.autograd.Functions over the C ABI (include/hpcs_mi355x.h).

Host-side glue only: shape checks, output allocation from torch's caching allocator, weight
re-layout, and the forward/backward call sequences.  All arithmetic runs in the gfx950 kernels.
"""
import torch

from . import _lib

call = _lib.call
VN_BN, VN_LRELU = 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# conv1 (two input channels) forms its pre-activations inside the point-wise kernels instead of materialising them
K2_ON_THE_FLY = __import__("os").environ.get("HPCS_K2", "1") != "0"
# HPCS_RC=1: edge-level EdgeConv layers (conv2 / conv4: 21 -> 21 channels on [B,21,3,N*k]) re-form their pre-activations on the
# matrix cores in every pass instead of materialising them (csrc/vn_recompute.hip: half the HBM traffic of that chain).  OPT-IN:
# measured slower in the step (10.13 against 9.42 ms) -- the chain is bound by the VALU cost of the VN point-wise arithmetic,
# which the recompute kernels pay at two waves per SIMD (profiles/r04_rc_experiment.md).  Tested equal to the streaming path.
RC_EDGE = __import__("os").environ.get("HPCS_RC", "0") == "1"
RC_MIN_L = 4096          # positions per cloud from which the recompute kernels are used (tests lower it)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.HpcsNativeError("hpcs_amd operators need HIP-device tensors (there is no CPU fallback)")


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _f32c_keep(W):
    """fp32 weight with a contiguous last dim (parameters already are: no copy, so column slices alias them)."""
    W = W if W.dtype == torch.float32 else W.float()
    return W if W.stride(-1) == 1 else W.contiguous()


class _ZeroArena:
    """fp64 accumulators (BatchNorm statistics and their backward reductions: ~26 per step) are slices of ONE buffer that
    the step driver clears with a single memset (`zero_arena_reset`), instead of one memset node per accumulator.
    Without a reset in the current step, or once the arena is used up, each accumulator is a fresh torch.zeros."""
    buf = None
    off = 0
    SIZE = 1 << 16


def zero_arena_reset(device, also=None):
    """Clears the arena; `also` (dp.FlatGradBucket): its flat gradient buffer is cleared by the same launch."""
    a = _ZeroArena
    if a.buf is None or a.buf.device != torch.device(device):
        a.buf = torch.empty(a.SIZE, dtype=torch.float64, device=device)
    if also is not None and also.flat.is_cuda:
        also.check_attached()
        call("hpcs_zero_buffers", a.buf, a.SIZE * 8, also.flat, also.numel * 4, None, 0, None, 0)
    else:
        a.buf.zero_()
        if also is not None:
            also.zero()
    a.off = 0


def zero_arena_release():
    _ZeroArena.off = _ZeroArena.SIZE          # nothing more is handed out until the next reset


def _zeros64(n, device):
    a = _ZeroArena
    if a.buf is not None and a.off + n <= a.SIZE and a.buf.device == device:
        v = a.buf[a.off:a.off + n]
        a.off += n
        return v
    return torch.zeros(n, dtype=torch.float64, device=device)


BN_FOLD = __import__("os").environ.get("HPCS_BN_FOLD", "1") != "0"     # A/B: 0 = statistics and finalisation as two launches
BN_FOLD_MAX = 1 << 20      # elements per channel up to which the fold is used: every workgroup of a statistics kernel pays a
#                            ticket (+9 us per kernel at 655 k elements: what the finalise launch cost; +20 us at 1.3 M: more)


def _bn_fold(count):
    return BN_FOLD and count <= BN_FOLD_MAX


def _zeros64_ticket(n, device):
    """n = 2C zeroed doubles + C zeroed 32-bit ticket counters behind them (the statistics kernels' last workgroup of a
    channel finalises it)."""
    nt = (n // 2 + 1) // 2
    buf = _zeros64(n + nt, device)
    return buf[:n], buf[n:].view(torch.int32)


def _pad4(n):
    return (n + 3) // 4 * 4


# ------------------------------------------------------------------------------------------------ raw GEMMs
class PrepCache:
    """Step-level batching of the weight re-layouts (one instance per HypHCStep).  Every PreparedW built while the cache is
    active is recorded with persistent operand buffers; from the next step on ONE multi-tensor launch at the head of the
    step refreshes all of them (`begin_step()`, valid until `end_step()` -- the optimiser then moves the weights) and
    PreparedW(...) finds its operands ready: 16 small launches per step less.  Outside a step nothing is cached or reused."""

    def __init__(self, parameters=()):
        # only weights that live inside a parameter's storage are recorded (a temporary operand has a new address every step)
        self.ranges = sorted((p.data_ptr(), p.data_ptr() + p.numel() * p.element_size()) for p in parameters)
        self.starts = [r[0] for r in self.ranges]
        self.fresh = False
        self.entries = {}          # key -> (At, Wp)
        self.order = []            # keys in table order
        self.table_dev = None
        self.dirty = False         # entries were added since the table was built

    def begin_step(self):
        """Called by the step driver before the forward pass: refreshes every recorded operand pair in one launch."""
        import struct
        global _ACTIVE_PREP
        _ACTIVE_PREP = self
        self.fresh = False
        if not self.order:
            return
        if self.dirty or self.table_dev is None:
            if torch.cuda.is_current_stream_capturing():
                return                                   # (no host->device copy inside a capture: per-layer launches this once)
            blob = bytearray()
            for key in self.order:
                At, Wp = self.entries[key]
                p1, ld1, M1, p2, ld2, M2, K = key
                blob += struct.pack("<4Q8i", p1, p2, At.data_ptr(), Wp.data_ptr(), ld1, M1, ld2, M2, K, At.shape[1], Wp.shape[1], 0)
            self.table_dev = torch.frombuffer(blob, dtype=torch.uint8).clone().to(self.entries[self.order[0]][0].device)
            self.dirty = False
        call("hpcs_prep_weights_multi", self.table_dev, len(self.order))
        self.fresh = True

    def owns(self, ptr):
        import bisect
        i = bisect.bisect_right(self.starts, ptr) - 1
        return i >= 0 and ptr < self.ranges[i][1]

    def end_step(self):
        global _ACTIVE_PREP
        self.fresh = False
        _ACTIVE_PREP = None


_ACTIVE_PREP = None


class PreparedW:
    """GEMM operands of W = [W1; W2] built by ONE kernel: At = W^T [K, pad4(M)] and Wp = W [M, pad4(K)]."""
    __slots__ = ("At", "Wp", "M", "K")

    def __init__(self, W1, W2=None):
        _need_cuda(W1, W2)
        if W1.stride(-1) != 1 or (W2 is not None and W2.stride(-1) != 1):
            W1 = W1.contiguous()
            W2 = None if W2 is None else W2.contiguous()
        W1 = W1 if W1.dtype == torch.float32 else W1.float()
        M1, K = W1.shape
        M2 = 0 if W2 is None else W2.shape[0]
        M = M1 + M2
        self.M, self.K = M, K
        c = _ACTIVE_PREP
        key = None
        if c is not None and (W2 is None or W2.dtype == torch.float32) and c.owns(W1.data_ptr()) and \
                (W2 is None or c.owns(W2.data_ptr())):
            key = (W1.data_ptr(), W1.stride(0), M1, 0 if W2 is None else W2.data_ptr(), 0 if W2 is None else W2.stride(0), M2, K)
            hit = c.entries.get(key) if c.fresh else None
            if hit is not None:
                self.At, self.Wp = hit                            # refreshed by this step's multi-tensor launch
                return
        self.At = torch.empty(K, _pad4(M), device=W1.device, dtype=torch.float32)
        self.Wp = torch.empty(M, _pad4(K), device=W1.device, dtype=torch.float32)
        call("hpcs_prep_weights", W1.data_ptr(), W1.stride(0), M1, None if W2 is None else W2.data_ptr(),
             0 if W2 is None else W2.stride(0), M2, K, self.At, self.At.shape[1], self.Wp, self.Wp.shape[1])
        if key is not None and key not in c.entries and not torch.cuda.is_current_stream_capturing():
            # (buffers allocated during a capture live in the graph's private pool: never adopted as persistent ones)
            c.entries[key] = (self.At, self.Wp)
            c.order.append(key)
            c.dirty = True


def _prep(W):
    return W if isinstance(W, PreparedW) else PreparedW(W)


def gemm_WX(W, X, nb, K, Nc, rowbias=None, out=None, accumulate=False):
    """out[b] (M x Nc) = W (M x K) @ X[b] (K x Nc); X / out are [nb, K, Nc] / [nb, M, Nc] views of contiguous data."""
    W = _prep(W)
    M = W.M
    if out is None:
        out = torch.empty(nb * M * Nc, device=X.device, dtype=torch.float32)
    call("hpcs_gemm_tn", W.At, W.At.shape[1], X, Nc, K * Nc, 0, out, Nc, M * Nc, 0, M, Nc, K, nb, 1, rowbias,
         1 if accumulate else 0)
    return out


def gemm_WtG(W, G, nb, M, Nc, out=None, accumulate=False):
    """out[b] (K x Nc) = W^T (K x M) @ G[b] (M x Nc)."""
    W = _prep(W)
    K = W.K
    if out is None:
        out = torch.empty(nb * K * Nc, device=G.device, dtype=torch.float32)
    call("hpcs_gemm_tn", W.Wp, W.Wp.shape[1], G, Nc, M * Nc, 0, out, Nc, K * Nc, 0, K, Nc, M, nb, 1, None,
         1 if accumulate else 0)
    return out


GRAPH_GEMM = __import__("os").environ.get("HPCS_GRAPH_GEMM", "1") != "0"   # A/B: 0 = materialised edge tensor + plain GEMM
NT_FUSED = __import__("os").environ.get("HPCS_NT_FUSED", "1") != "0"     # A/B: 0 = split partials summed by follow-up launches


def _nt_call(G, ldg, sGo, X, ldx, sXo, M, K, Nc, nb, dst_ptr, lddw, accumulate, device):
    """hpcs_gemm_nt with its split reduction inside the launch (tickets from the step's zeroed arena), or -- HPCS_NT_FUSED=0 --
    followed by the split_reduce launches."""
    nbytes = call("hpcs_gemm_nt_workspace", M, K, Nc, nb, 1)
    ws = torch.empty(nbytes // 4 + 4, device=device, dtype=torch.float32)
    if NT_FUSED:
        nt = call("hpcs_gemm_nt_tickets", M, K, Nc, nb, 1)
        tickets = _zeros64((nt + 1) // 2, device)             # 32-bit words, zero; the kernel leaves them zero
        call("hpcs_gemm_nt_fused", G, ldg, sGo, 0, X, ldx, sXo, 0, M, K, Nc, nb, 1, dst_ptr, lddw, accumulate, ws, nbytes,
             tickets, nt)
    else:
        call("hpcs_gemm_nt", G, ldg, sGo, 0, X, ldx, sXo, 0, M, K, Nc, nb, 1, dst_ptr, lddw, accumulate, ws, nbytes)


def gemm_GXt(G, X, nb, M, K, Nc, out=None):
    """dW (M x K) = sum_b G[b] (M x Nc) @ X[b]^T (K x Nc).  out: a [M, K] view with unit column stride that the result is
    ADDED into (a slice of the flat gradient bucket: no temporary, no separate add)."""
    if out is not None:
        if tuple(out.shape) != (M, K) or out.stride(1) != 1:
            raise ValueError("gemm_GXt: out must be an [M, K] view with contiguous rows")
        _nt_call(G, Nc, M * Nc, X, Nc, K * Nc, M, K, Nc, nb, out.data_ptr(), out.stride(0), 1, G.device)
        return out
    dW = torch.empty(M, K, device=G.device, dtype=torch.float32)
    _nt_call(G, Nc, M * Nc, X, Nc, K * Nc, M, K, Nc, nb, dW.data_ptr(), K, 0, G.device)
    return dW


def _stacked_sink(sink_f, sink_d, Cout, Cd, Cin):
    """One [Cout+Cd, Cin] view over the two gradient slots when they lie back to back in the flat bucket (dp.py lays the
    feature / direction weights of a VN layer out that way), else None."""
    if sink_f is None or tuple(sink_f.shape) != (Cout, Cin) or not sink_f.is_contiguous():
        return None
    if Cd == 0:
        return sink_f
    if sink_d is None or tuple(sink_d.shape) != (Cd, Cin) or not sink_d.is_contiguous():
        return None
    if sink_d.data_ptr() != sink_f.data_ptr() + 4 * Cout * Cin or sink_d.untyped_storage().data_ptr() != sink_f.untyped_storage().data_ptr():
        return None
    return torch.as_strided(sink_f, (Cout + Cd, Cin), (Cin, 1))


# ------------------------------------------------------------------------------------------------ side-stream dW
class _SideWGrad:
    """Weight-gradient GEMMs are off the backward critical path (nothing consumes dW before the optimiser), so inside
    `deferred_weight_grads()` they run on a second HIP stream and add straight into the parameter's `.grad` (the flat
    bucket of dp.py) while the main stream continues with the data-gradient chain: the MFMA-bound dW kernels overlap
    the HBM-bound element-wise passes.  Outside the context (or for parameters without a pre-allocated `.grad`)
    dW is returned to autograd as usual."""
    enabled = False
    forward = False
    stream = None
    used = False
    keep = []
    fwd_pending = False          # forward work was forked onto the side stream and not joined yet


class forward_side_stream:
    """Context manager (step driver): inside it the graph layers build their transposed kNN graph for the backward pass
    already in the forward pass, on the side stream, where it overlaps the forward kernels (45 us per layer off the
    backward's critical path).  Only where a backward pass is certain to follow: the side stream is joined there."""

    def __enter__(self):
        if _SideWGrad.stream is None:
            _SideWGrad.stream = torch.cuda.Stream()
        _SideWGrad.forward = True
        return self

    def __exit__(self, *exc):
        join_forward_side()
        _SideWGrad.forward = False
        return False


def fork_forward_side():
    """Inside forward_side_stream(): the side stream, made to wait for everything enqueued on the current stream so far, for
    forward work that nothing on the main stream needs until `join_forward_side()` (the CosFace branch of the loss, which runs
    beside the sampled HypHC kernel).  None outside that context (the caller then launches in line).  Outputs must be
    allocated BEFORE switching streams (on the current stream), so that their blocks obey the main stream's ordering."""
    if not _SideWGrad.forward or _SideWGrad.stream is None:
        return None
    _SideWGrad.stream.wait_stream(torch.cuda.current_stream())
    _SideWGrad.fwd_pending = True
    return _SideWGrad.stream


def join_forward_side():
    if _SideWGrad.fwd_pending:
        torch.cuda.current_stream().wait_stream(_SideWGrad.stream)
        _SideWGrad.fwd_pending = False


class deferred_weight_grads:
    """Context manager around `loss.backward()`; on exit the main stream waits for the side stream."""

    def __enter__(self):
        if _SideWGrad.stream is None:
            _SideWGrad.stream = torch.cuda.Stream()
        _SideWGrad.enabled = True
        return self

    def __exit__(self, *exc):
        _SideWGrad.enabled = False
        if _SideWGrad.used:
            torch.cuda.current_stream().wait_stream(_SideWGrad.stream)
            _SideWGrad.used = False
        _SideWGrad.keep.clear()
        return False


def _grad_sink(spec):
    """spec = (parameter, index) -> the view of parameter.grad that dW accumulates into, or None."""
    if not _SideWGrad.enabled or spec is None:
        return None
    p, index = spec
    if not (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_cuda):
        return None
    return p.grad[index]


def _bn_sinks(ctx_params, needs_gamma, needs_beta):
    """(ggamma_view, gbeta_view) when both BatchNorm parameters take their gradients straight into .grad (inside
    deferred_weight_grads(): the backward-apply kernels then add them there themselves), else (None, None)."""
    gamma, beta = ctx_params
    if not (needs_gamma and needs_beta) or gamma is None or beta is None:
        return None, None
    sg, sb = _grad_sink((gamma, ...)), _grad_sink((beta, ...))
    if sg is None or sb is None or not (sg.is_contiguous() and sb.is_contiguous()):
        return None, None
    return sg, sb


def _bn_grads(ctx_params, red, C, needs_gamma, needs_beta):
    """(ggamma, gbeta) from the fp64 reduction red = [sum_beta(C) | sum_gamma(C)]: added into .grad on the side stream
    inside deferred_weight_grads(), else returned to autograd as fp32."""
    gamma, beta = ctx_params
    ggamma = gbeta = None
    sg = _grad_sink((gamma, ...)) if (needs_gamma and gamma is not None) else None
    sb = _grad_sink((beta, ...)) if (needs_beta and beta is not None) else None
    if sg is not None and sb is not None:
        def work():
            sg.add_(red[C:])
            sb.add_(red[:C])
        _on_side_stream(work, (red,))
    else:
        ggamma = red[C:].float() if needs_gamma else None
        gbeta = red[:C].float() if needs_beta else None
    return ggamma, gbeta


SIDE_INLINE_BIG = int(__import__("os").environ.get("HPCS_SIDE_INLINE_BIG", "0"))    # experiment: 0 = everything on the side stream


def _on_side_stream(work, keep_alive, size=0):
    if SIDE_INLINE_BIG and size >= SIDE_INLINE_BIG:
        work()                                   # in line, on the main stream
        return
    main = torch.cuda.current_stream()
    side = _SideWGrad.stream
    side.wait_stream(main)
    with torch.cuda.stream(side):
        work()
    # operands were allocated on the main stream and are read on the side stream: keep them referenced until the main
    # stream has waited for the side stream (__exit__), so the caching allocator cannot hand their blocks out earlier
    # (cheaper than Tensor.record_stream, whose deferred frees made the allocator fall back to hipMalloc/hipFree)
    _SideWGrad.keep.append(keep_alive)
    _SideWGrad.used = True


# ------------------------------------------------------------------------------------------------ a1 / a2
def knn(x, k):
    _need_cuda(x)
    if x.dim() != 3:
        raise ValueError("knn expects x of shape [B, D, N]")
    x = _f32c(x.detach())
    B, D, N = x.shape
    if not 0 < k <= N:
        raise ValueError(f"k={k} must be in 1..N={N}")
    idx = torch.empty(B, N, k, dtype=torch.int64, device=x.device)
    nbytes = call("hpcs_knn_workspace", B, N)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    call("hpcs_knn", x, B, D, N, k, idx, ws, nbytes, tag=(B, D, N, k))
    return idx


class EdgeGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        B, C, three, N = x.shape
        k = idx.shape[-1]
        x = _f32c(x)
        idx = idx.contiguous()
        out = torch.empty(B, 2 * C, 3, N, k, device=x.device, dtype=torch.float32)
        call("hpcs_edge_gather_fwd", x, idx, B, C, N, k, out)
        ctx.save_for_backward(idx)
        ctx.dims = (B, C, N, k)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, C, N, k = ctx.dims
        gx = torch.empty(B, C, 3, N, device=g.device, dtype=torch.float32)
        nbytes = call("hpcs_edge_gather_bwd_workspace", B, C, N)
        ws = torch.empty(nbytes // 4, device=g.device, dtype=torch.float32)
        call("hpcs_edge_gather_bwd", _f32c(g), idx, B, C, N, k, gx, ws, nbytes)
        return gx, None


# ------------------------------------------------------------------------------------------------ channel mix
class ChanMixFn(torch.autograd.Function):
    """y[b,o,...] = sum_c W[o,c] x[b,c,...] (+ rowbias[b,o])  (nn.Linear over dim 1; Conv1d k=1)."""

    @staticmethod
    def forward(ctx, x, W, rowbias=None, sink=None):
        ctx.sink = sink
        x = _f32c(x)
        W = _f32c(W)
        B, K = x.shape[0], x.shape[1]
        Nc = x[0, 0].numel()
        M = W.shape[0]
        rb = _f32c(rowbias) if rowbias is not None else None
        Wq = PreparedW(W)
        y = gemm_WX(Wq, x, B, K, Nc, rowbias=rb).view(B, M, *x.shape[2:])
        ctx.save_for_backward(x, Wq.Wp)
        ctx.mk = (M, K)
        ctx.has_bias = rowbias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, Wp = ctx.saved_tensors
        g = _f32c(g)
        B = x.shape[0]
        Nc = x[0, 0].numel()
        M, K = ctx.mk
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B * K * Nc, device=g.device)
            call("hpcs_gemm_tn", Wp, Wp.shape[1], g, Nc, M * Nc, 0, gx, Nc, K * Nc, 0, K, Nc, M, B, 1, None, 0)
            gx = gx.view_as(x)
        if ctx.needs_input_grad[1]:
            dst = _grad_sink(ctx.sink)
            if dst is not None and dst.dim() == 2 and dst.stride(1) == 1:
                _on_side_stream(lambda: gemm_GXt(g, x, B, M, K, Nc, out=dst), (g, x))     # straight into the bucket
            elif dst is not None:
                _on_side_stream(lambda: dst.add_(gemm_GXt(g, x, B, M, K, Nc)), (g, x))
            else:
                gW = gemm_GXt(g, x, B, M, K, Nc)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(B, M, device=g.device)
            call("hpcs_rowsum", g, B * M, Nc, 1.0, gb)
        return gx, gW, gb, None


def chan_mix(x, W, rowbias=None, sink=None):
    """sink = (parameter, index) with parameter[index] == W: lets the backward add dW into parameter.grad[index] on the
    side stream (see deferred_weight_grads)."""
    _need_cuda(x, W)
    return ChanMixFn.apply(x, W, rowbias, sink)


# ------------------------------------------------------------------------------------------------ row-vector linear
def _nt_into(G, ldg, X, ldx, R, M, K, out, ldo, accumulate=False):
    """out[R, M] (row stride ldo) = G[R, K] @ X[M, K]^T, reduction over the contiguous index (split over workgroups)."""
    _nt_call(G.data_ptr(), ldg, 0, X.data_ptr(), ldx, 0, R, M, K, 1, out.data_ptr(), ldo, 1 if accumulate else 0, G.device)
    return out


def _rowmajor_view(W):
    """(tensor, row stride) for a 2-D weight view whose last dim is contiguous (column slices stay un-copied)."""
    if W.dim() != 2 or W.stride(1) != 1:
        W = W.contiguous()
    return W, W.stride(0)


class RowLinearFn(torch.autograd.Function):
    """out[R, M] = vec[R, K] @ W[M, K]^T for a FEW rows R (one per cloud): the N-constant channels of conv8 enter as a
    [B,256] row bias (vn_dgcnn_partseg.py:91-96).  As a per-cloud channel mix this is 32 GEMMs with ONE column each; in
    this orientation it is a single split-reduction GEMM, and both gradients are single GEMMs too."""

    @staticmethod
    def forward(ctx, vec, W, sink=None):
        vec = _f32c(vec)
        W, ldw = _rowmajor_view(W if W.dtype == torch.float32 else W.float())
        R, K = vec.shape
        M = W.shape[0]
        out = torch.empty(R, M, device=vec.device, dtype=torch.float32)
        _nt_into(vec, K, W, ldw, R, M, K, out, M)
        ctx.save_for_backward(vec, W)
        ctx.sink = sink
        return out

    @staticmethod
    def backward(ctx, g):
        vec, W = ctx.saved_tensors
        g = _f32c(g)
        R, K = vec.shape
        M, ldw = W.shape[0], W.stride(0)
        gvec = gW = None
        if ctx.needs_input_grad[0]:
            gT = g.t().contiguous()                                   # [M, R] = At of the data-gradient GEMM
            gvec = torch.empty(R, K, device=g.device, dtype=torch.float32)
            call("hpcs_gemm_tn", gT, R, W.data_ptr(), ldw, 0, 0, gvec, K, 0, 0, R, K, M, 1, 1, None, 0)
        if ctx.needs_input_grad[1]:
            dst = _grad_sink(ctx.sink)
            if dst is not None and dst.stride(-1) == 1:
                def work():             # dW[M, K] += g^T vec, straight into the parameter's gradient
                    call("hpcs_gemm_tn", g, M, vec, K, 0, 0, dst.data_ptr(), dst.stride(0), 0, 0, M, K, R, 1, 1, None, 1)
                _on_side_stream(work, (g, vec))
            else:
                gW = torch.empty(M, K, device=g.device, dtype=torch.float32)
                call("hpcs_gemm_tn", g, M, vec, K, 0, 0, gW, K, 0, 0, M, K, R, 1, 1, None, 0)
        return gvec, gW, None


def row_linear(vec, W, sink=None):
    _need_cuda(vec, W)
    return RowLinearFn.apply(vec, W, sink)


# ------------------------------------------------------------------------------------------------ a3
class GradHub:
    """Hands the SPARSE part of a tensor's gradient to the Function that produces its DENSE part.

    g6 = conv6(x123) feeds vn1 (dense gradient, a GEMM) and the fused project+max (gradient only at the arg-max
    positions).  Through autograd the sparse part costs a zero-filled [B,341,3,N] tensor plus an add over it; with a
    hub the later node (ProjectMaxFn, consumer of vn1's frame) queues its scatter and the earlier node (VNLayerFn of
    vn1, whose backward therefore runs AFTER it) applies it to its freshly written gx.  Only valid when `taker` is an
    ancestor of `giver` in the graph, which VN_DGCNN_partseg guarantees by construction; a giver that runs after the
    taker raises instead of dropping a gradient."""

    def __init__(self):
        self.pending = []
        self.taken = False

    def give(self, fn):
        if self.taken:
            raise RuntimeError("GradHub: the dense producer ran before the sparse one (graph order violated)")
        self.pending.append(fn)

    def take(self, gx):
        self.taken = True
        for fn in self.pending:
            fn(gx)
        self.pending.clear()


def _gemm_inner3(W, x, B, K, L, rowbias3):
    """PD[b,:,a,:] = W @ x[b,:,a,:] + rowbias3[b,a,:]  -- walks a [B,K,3,L] tensor as 3B matrices of L columns."""
    W = _prep(W)
    M = W.M
    out = torch.empty(B * M * 3 * L, device=x.device, dtype=torch.float32)
    call("hpcs_gemm_tn", W.At, W.At.shape[1], x, 3 * L, K * 3 * L, L, out, 3 * L, M * 3 * L, L, M, L, K, B, 3, rowbias3, 0)
    return out


class VNLayerFn(torch.autograd.Function):
    """p = Wf x ; [VNBatchNorm] ; [VN-LeakyReLU with d = Wd x] ; [mean over pool_k consecutive positions].

    x [B,Cin,3,*S] -> [B,Cout,3,*S] (or [B,Cout,3,S0] when pool_k = S1 > 0).
    One stacked GEMM produces PD = [Wf;Wd] x; backward recomputes the point-wise state from PD.
    xconst [B,Cc,3] (optional): Cc further input channels that are constant over the positions (the weights' last Cc
    columns); they enter as a per-(cloud, xyz) row bias instead of being expanded (vn_dgcnn_partseg.py:82-83).
    """

    @staticmethod
    def forward(ctx, x, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool_k, flags, xconst, hub=None):
        ctx.wparams = (Wf, Wd)
        ctx.hub = hub
        ctx.bnparams = (gamma, beta)
        x = _f32c(x)
        B, Cin = x.shape[0], x.shape[1]
        L = x[0, 0, 0].numel()
        Cout = Wf.shape[0]
        Cd = Wd.shape[0] if (Wd is not None and (flags & VN_LRELU)) else 0
        Wd_ = Wd if Cd else None
        Ctot = Cout + Cd
        Cc = 0
        xcT = None
        ctx.wversions = (Wf._version, None if Wd is None else Wd._version)
        ctx.mean_mode = isinstance(xconst, str)
        if ctx.mean_mode:
            # the constant channels ARE the mean of x over the positions (std_feature's vn1, vn_dgcnn_partseg.py:82-83):
            # computed here and returned as a second output, so that in backward everything that flows into the mean
            # re-enters gx as a row bias of the data-gradient GEMM instead of a broadcast tensor + an autograd add
            if xconst != "mean":
                raise ValueError("xconst must be a tensor, None or 'mean'")
            xconst = torch.empty(B, Cin, 3, device=x.device, dtype=torch.float32)
            call("hpcs_rowsum", x, B * Cin * 3, L, 1.0 / L, xconst)
        ctx.rc = bool(RC_EDGE and xconst is None and Cd == Cout and L >= RC_MIN_L and
                      call("hpcs_vn_rc_supported", Cin, Cout, Cd, L, pool_k, flags))
        if ctx.rc:
            # no pre-activation tensor: statistics and apply (+ mean over k) each re-form their PD tile from x
            WxP = PreparedW(Wf, Wd_)
            Wx = WxP.Wp
            dev = x.device
            mean = torch.empty(Cout, device=dev)
            rstd = torch.empty(Cout, device=dev)
            if training:
                sums, ticket = _zeros64_ticket(2 * Cout, dev)
                call("hpcs_vn_rc_stats", x, Wx, Wx.shape[1], B, Cin, Cout, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd)
            else:
                call("hpcs_bn_finalize", None, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 0)
            out = torch.empty((B, Cout, 3, L // pool_k) if pool_k else (B, Cout, *x.shape[2:]), device=dev)
            call("hpcs_vn_rc_apply_fwd", x, Wx, Wx.shape[1], mean, rstd, gamma, beta, B, Cin, Cout, L, pool_k, slope, out)
            ctx.save_for_backward(x, Wx, None, None, mean, rstd, gamma, beta)
            ctx.cfg = (B, Cin, 0, Cout, Cd, L, pool_k, slope, flags, bool(training))
            return out
        if xconst is not None:
            xconst = _f32c(xconst)
            Cc = xconst.shape[1]
            WxP = PreparedW(Wf[:, :Cin], None if Wd_ is None else Wd_[:, :Cin])      # column slices: no copies
            # bias[(b,xyz), :] = [Wf_c; Wd_c] xconst[b, :, xyz]: ONE split-reduction GEMM over the 3B rows per weight
            # block (as a per-cloud channel mix it would be B GEMMs of 3 columns each)
            xcT = xconst.permute(0, 2, 1).contiguous()                                # [B,3,Cc]
            bias = torch.empty(B, 3, Ctot, device=x.device, dtype=torch.float32)
            Wfc, ldf = _rowmajor_view(_f32c_keep(Wf)[:, Cin:])
            _nt_into(xcT, Cc, Wfc, ldf, 3 * B, Cout, Cc, bias, Ctot)
            if Cd:
                Wdc, ldd = _rowmajor_view(_f32c_keep(Wd_)[:, Cin:])
                _nt_into(xcT, Cc, Wdc, ldd, 3 * B, Cd, Cc, bias[:, :, Cout:], Ctot)
            PD = _gemm_inner3(WxP, x, B, Cin, L, bias)
        else:
            WxP = PreparedW(Wf, Wd_)
            PD = gemm_WX(WxP, x, B, Cin, 3 * L)
        Wx = WxP.Wp
        dev = x.device
        mean = rstd = None
        if flags & VN_BN:
            mean = torch.empty(Cout, device=dev)
            rstd = torch.empty(Cout, device=dev)
            if training and not _bn_fold(B * L):
                sums = _zeros64(2 * Cout, dev)
                call("hpcs_vn_norm_stats", PD, B, Cout, Ctot, L, sums, 1)
                call("hpcs_bn_finalize", sums, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1)
            elif training:
                sums, ticket = _zeros64_ticket(2 * Cout, dev)
                call("hpcs_vn_norm_stats_finalize", PD, B, Cout, Ctot, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd)
            else:
                call("hpcs_bn_finalize", None, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 0)
        if pool_k:
            out = torch.empty(B, Cout, 3, L // pool_k, device=dev)
        else:
            out = torch.empty(B, Cout, *x.shape[2:], device=dev)
        call("hpcs_vn_apply_fwd", PD, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, out)
        ctx.save_for_backward(x, Wx, xcT, PD, mean, rstd, gamma, beta)
        ctx.cfg = (B, Cin, Cc, Cout, Cd, L, pool_k, slope, flags, bool(training))
        if ctx.mean_mode:
            return out, xconst
        return out

    @staticmethod
    def backward(ctx, g, g_mean=None):
        x, Wx, xcT, PD, mean, rstd, gamma, beta = ctx.saved_tensors
        B, Cin, Cc, Cout, Cd, L, pool_k, slope, flags, training = ctx.cfg
        g = _f32c(g)
        dev = g.device
        Ctot = Cout + Cd
        red = None
        gx = gW = gWd = ggamma = gbeta = gxc = None
        if ctx.rc:
            # backward reduction and apply re-form PD from x; the apply pass writes GPD once (for the weight-gradient GEMM
            # below, on the side stream) and forms gx = W^T GPD from the same registers: no edge-level data-gradient GEMM
            red = _zeros64(2 * Cout, dev)
            call("hpcs_vn_rc_bwd_reduce", x, Wx, Wx.shape[1], g, mean, rstd, gamma, beta, B, Cin, Cout, L, pool_k, slope, red)
            GPD = torch.empty(B * Ctot * 3 * L, device=dev)
            gx = torch.empty_like(x)
            sg, sb = _bn_sinks(ctx.bnparams, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
            call("hpcs_vn_rc_bwd_apply", x, Wx, Wx.shape[1], g, mean, rstd, gamma, beta, red, 1 if training else 0, B, Cin, Cout, L,
                 pool_k, slope, GPD, gx, sg, sb)
            if ctx.hub is not None:
                ctx.hub.take(gx)
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                Wf_, Wd_ = ctx.wparams
                sink_f = _grad_sink((Wf_, ...)) if ctx.needs_input_grad[1] else None
                sink_d = _grad_sink((Wd_, ...)) if ctx.needs_input_grad[2] else None
                both = _stacked_sink(sink_f, sink_d, Cout, Cd, Cin)
                if both is not None and sink_d is not None:
                    _on_side_stream(lambda: gemm_GXt(GPD, x, B, Ctot, Cin, 3 * L, out=both), (GPD, x), size=Ctot * Cin)
                elif (sink_f is not None or not ctx.needs_input_grad[1]) and (sink_d is not None or not ctx.needs_input_grad[2]):
                    def work():
                        gWcat = gemm_GXt(GPD, x, B, Ctot, Cin, 3 * L)
                        if sink_f is not None:
                            sink_f.add_(gWcat[:Cout])
                        if sink_d is not None:
                            sink_d.add_(gWcat[Cout:])
                    _on_side_stream(work, (GPD, x))
                else:
                    gWcat = gemm_GXt(GPD, x, B, Ctot, Cin, 3 * L)
                    gW, gWd = gWcat[:Cout], gWcat[Cout:]
            if sg is None:
                ggamma, gbeta = _bn_grads(ctx.bnparams, red, Cout, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
            return gx, gW, gWd, ggamma, gbeta, None, None, None, None, None, None, None, None
        if flags & VN_BN:
            red = _zeros64(2 * Cout, dev)
            call("hpcs_vn_bwd_reduce", PD, g, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, red, 1)
        GPD = torch.empty(B * Ctot * 3 * L, device=dev)
        sg, sb = _bn_sinks(ctx.bnparams, ctx.needs_input_grad[3], ctx.needs_input_grad[4]) if (flags & VN_BN) else (None, None)
        call("hpcs_vn_bwd_apply", PD, g, mean, rstd, gamma, beta, red, 1 if training else 0, B, Cout, Cd, L, pool_k, slope,
             flags, GPD, sg, sb)
        ST = None
        if Cc:
            S = torch.empty(B, Ctot, 3, device=dev)
            call("hpcs_rowsum", GPD, B * Ctot * 3, L, 1.0, S)
            ST = S.permute(0, 2, 1).contiguous()                              # [(b,xyz), Ctot]
            if ctx.needs_input_grad[11] or (ctx.mean_mode and ctx.needs_input_grad[0]):
                # gxc[(b,xyz), :] = ST [Wf_c; Wd_c]: two accumulating GEMMs (one per weight block), At = ST^T
                Wf_, Wd_ = ctx.wparams          # the constant-channel columns are read from the live weights
                if (Wf_._version, None if Wd_ is None else Wd_._version) != ctx.wversions:
                    raise RuntimeError("VNLayerFn: weights were modified in place between forward and backward")
                STt = ST.view(3 * B, Ctot).t().contiguous()                   # [Ctot, 3B]
                gxcT = torch.empty(3 * B, Cc, device=dev)
                Wfc, ldf = _rowmajor_view(_f32c_keep(Wf_)[:, Cin:])
                call("hpcs_gemm_tn", STt, 3 * B, Wfc.data_ptr(), ldf, 0, 0, gxcT, Cc, 0, 0, 3 * B, Cc, Cout, 1, 1, None, 0)
                if Cd:
                    Wdc, ldd = _rowmajor_view(_f32c_keep(Wd_)[:, Cin:])
                    call("hpcs_gemm_tn", STt[Cout:], 3 * B, Wdc.data_ptr(), ldd, 0, 0, gxcT, Cc, 0, 0, 3 * B, Cc, Cd, 1, 1,
                         None, 1)
                gxc = gxcT.view(B, 3, Cc).permute(0, 2, 1)
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B * Cin * 3 * L, device=dev)
            if ctx.mean_mode:
                # d mean / d x = 1/L on every position: a per-(cloud, xyz, channel) row bias of the data-gradient GEMM
                gm = gxcT.view(B, 3, Cin)
                if g_mean is not None:
                    gm = gm + _f32c(g_mean).permute(0, 2, 1)
                gm = (gm * (1.0 / L)).contiguous()
                call("hpcs_gemm_tn", Wx, Wx.shape[1], GPD, 3 * L, Ctot * 3 * L, L, gx, 3 * L, Cin * 3 * L, L, Cin, L, Ctot, B, 3,
                     gm, 0)
                gxc = None
            else:
                call("hpcs_gemm_tn", Wx, Wx.shape[1], GPD, 3 * L, Ctot * 3 * L, 0, gx, 3 * L, Cin * 3 * L, 0, Cin, 3 * L, Ctot,
                     B, 1, None, 0)
            gx = gx.view_as(x)
            if ctx.hub is not None:
                ctx.hub.take(gx)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            def weight_grads():
                gWcat = gemm_GXt(GPD, x, B, Ctot, Cin, 3 * L)
                if Cc:                  # dWc[Ctot, Cc] = ST^T xcT: reduction over the 3B rows
                    gWc = torch.empty(Ctot, Cc, device=dev)
                    call("hpcs_gemm_tn", ST, Ctot, xcT, Cc, 0, 0, gWc, Cc, 0, 0, Ctot, Cc, 3 * B, 1, 1, None, 0)
                    gWcat = torch.cat([gWcat, gWc], dim=1)
                return gWcat

            Wf_, Wd_ = ctx.wparams
            sink_f = _grad_sink((Wf_, ...)) if ctx.needs_input_grad[1] else None
            sink_d = _grad_sink((Wd_, ...)) if (Cd and ctx.needs_input_grad[2]) else None
            both = None if Cc else _stacked_sink(sink_f, sink_d, Cout, Cd, Cin)
            if both is not None and (sink_d is not None or not Cd):
                # feature and direction weights lie back to back in the bucket: the stacked GEMM adds straight into them
                _on_side_stream(lambda: gemm_GXt(GPD, x, B, Ctot, Cin, 3 * L, out=both), (GPD, x), size=Ctot * Cin)
            elif (sink_f is not None or not ctx.needs_input_grad[1]) and \
                    (sink_d is not None or not (Cd and ctx.needs_input_grad[2])):
                def work():
                    gWcat = weight_grads()
                    if sink_f is not None:
                        sink_f.add_(gWcat[:Cout])
                    if sink_d is not None:
                        sink_d.add_(gWcat[Cout:])
                _on_side_stream(work, (GPD, x, ST, xcT))
            else:
                gWcat = weight_grads()
                gW = gWcat[:Cout]
                gWd = gWcat[Cout:] if Cd else None
        if (flags & VN_BN) and sg is None:
            ggamma, gbeta = _bn_grads(ctx.bnparams, red, Cout, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        return gx, gW, gWd, ggamma, gbeta, None, None, None, None, None, None, gxc, None


def vn_layer(x, Wf, Wd, gamma, beta, rmean, rvar, training, slope=0.2, pool_k=0, flags=VN_BN | VN_LRELU, xconst=None,
             hub=None):
    _need_cuda(x, Wf)
    return VNLayerFn.apply(x, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool_k, flags, xconst, hub)


class VNEdgeLayerFn(torch.autograd.Function):
    """get_graph_feature + VNLinearLeakyReLU (+ mean over k) on point features, without any N*k-sized GEMM operand.

    feat [B,C,3,N], idx [B,N,k] -> [B,Cout,3,N,k] (or [B,Cout,3,N] with pool).  W = [Wf;Wd] has 2C columns
    (difference part | centre part); W [x_nbr-x_ctr; x_ctr] = W_a x_nbr + (W_b-W_a) x_ctr, so one POINT-level GEMM
    gives UV = [W_a; W_b-W_a] feat and the edge kernels gather from it (include/hpcs_mi355x.h, "gather layers").
    """

    @staticmethod
    def forward(ctx, feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool):
        feat = _f32c(feat)
        idx = idx.contiguous()
        B, C, _, N = feat.shape
        k = idx.shape[-1]
        Cout, Cd = Wf.shape[0], Wd.shape[0]
        Ctot = Cout + Cd
        Wcat = _f32c(torch.cat([Wf, Wd], 0))                       # [Ctot, 2C]
        Wa, Wb = Wcat[:, :C], Wcat[:, C:]
        Wuv = torch.cat([Wa, Wb - Wa], 0).contiguous()             # [2Ctot, C]
        UV = gemm_WX(Wuv, feat, B, C, 3 * N)                       # [B, 2Ctot, 3, N]
        dev = feat.device
        mean = torch.empty(Cout, device=dev)
        rstd = torch.empty(Cout, device=dev)
        sums = None
        if training:
            sums = _zeros64(2 * Cout, dev)
            call("hpcs_vn_edge_norm_stats", UV, idx, B, Cout, Ctot, N, k, sums, 1)
        call("hpcs_bn_finalize", sums, Cout, B * N * k, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1 if training else 0)
        out = torch.empty((B, Cout, 3, N) if pool else (B, Cout, 3, N, k), device=dev)
        flags = VN_BN | VN_LRELU
        call("hpcs_vn_edge_apply_fwd", UV, idx, mean, rstd, gamma, beta, B, Cout, Cd, N, k, 1 if pool else 0, slope, flags, out)
        ctx.save_for_backward(feat, idx, Wuv, UV, mean, rstd, gamma, beta)
        ctx.cfg = (B, C, N, k, Cout, Cd, bool(pool), slope, flags, bool(training))
        return out

    @staticmethod
    def backward(ctx, g):
        feat, idx, Wuv, UV, mean, rstd, gamma, beta = ctx.saved_tensors
        B, C, N, k, Cout, Cd, pool, slope, flags, training = ctx.cfg
        Ctot = Cout + Cd
        g = _f32c(g)
        dev = g.device
        pl = 1 if pool else 0
        red = _zeros64(2 * Cout, dev)
        call("hpcs_vn_edge_bwd_reduce", UV, idx, g, mean, rstd, gamma, beta, B, Cout, Cd, N, k, pl, slope, flags, red, 1)
        GPD = torch.empty(B * Ctot * 3 * N * k, device=dev)
        call("hpcs_vn_edge_bwd_apply", UV, idx, g, mean, rstd, gamma, beta, red, 1 if training else 0, B, Cout, Cd, N, k, pl,
             slope, flags, GPD)
        gUV = torch.empty(B * 2 * Ctot * 3 * N, device=dev)
        nbytes = call("hpcs_edge_grad_reduce_workspace", B, Ctot, N)
        ws = torch.empty(nbytes // 4, device=dev)
        call("hpcs_edge_grad_reduce", GPD, idx, B, Ctot, N, k, gUV, ws, nbytes)
        del GPD
        gfeat = gWf = gWd = None
        if ctx.needs_input_grad[0]:
            gfeat = gemm_WtG(Wuv, gUV, B, 2 * Ctot, 3 * N).view_as(feat)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gWuv = gemm_GXt(gUV, feat, B, 2 * Ctot, C, 3 * N)                  # [2Ctot, C]
            gWa = gWuv[:Ctot] - gWuv[Ctot:]
            gWcat = torch.cat([gWa, gWuv[Ctot:]], dim=1)                       # [Ctot, 2C]
            gWf, gWd = gWcat[:Cout], gWcat[Cout:]
        return gfeat, None, gWf, gWd, red[Cout:].float(), red[:Cout].float(), None, None, None, None, None


class VNGraphLayerFn(torch.autograd.Function):
    """get_graph_feature + VNLinearLeakyReLU (+ mean over k), forward EXACTLY as the two separate ops (the edge tensor
    [B,2C,3,N,k] is built and the channel mix runs on it: the reference's rounding sequence, difference before the mix),
    backward at POINT level: the gradient of the stacked GEMM output is reduced over the graph first
    (hpcs_edge_grad_reduce: gU by the transposed adjacency, gV over the k outgoing edges), then

        g feat = Wa^T gU + (Wb - Wa)^T gV          dWa = gU feat^T - gV feat^T          dWb = gV feat^T

    (W = [Wf;Wd] = [Wa | Wb], difference | centre columns).  This replaces the edge-level data-gradient GEMM (330 MB in,
    330 MB out), the scatter of its output and the edge-level weight-gradient GEMM by GEMMs k times smaller, and the edge
    tensor is not kept for backward.
    """

    @staticmethod
    def forward(ctx, feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool_k):
        ctx.wparams = (Wf, Wd)
        ctx.bnparams = (gamma, beta)
        feat = _f32c(feat)
        idx = idx.contiguous()
        B, C, _, N = feat.shape
        k = idx.shape[-1]
        L = N * k
        Cout, Cd = Wf.shape[0], Wd.shape[0]
        Ctot = Cout + Cd
        dev = feat.device
        WxP = PreparedW(Wf, Wd)
        flags = VN_BN | VN_LRELU
        mean = torch.empty(Cout, device=dev)
        rstd = torch.empty(Cout, device=dev)
        sums, ticket = _zeros64_ticket(2 * Cout, dev) if training else (None, None)
        k2 = C == 1 and K2_ON_THE_FLY
        # the channel mix with the edge tensor formed in its operand loader (hpcs_gemm_tn_graph: bit-identical product, no
        # 330 MB edge tensor written and read back); otherwise the edge tensor is materialised for the plain GEMM
        in_loader = GRAPH_GEMM and not k2 and Ctot <= 64 and 2 * C <= 64 and k % 4 == 0 and idx.dtype == torch.int64
        E = None
        if not in_loader:
            E = torch.empty(B, 2 * C, 3, N, k, device=dev, dtype=torch.float32)
            call("hpcs_edge_gather_fwd", feat, idx, B, C, N, k, E)
        if in_loader:
            Wm = _prep(WxP)
            PD = torch.empty(B * Ctot * 3 * L, device=dev, dtype=torch.float32)
            nbytes = call("hpcs_gemm_tn_graph_workspace", B, C, N)
            ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
            call("hpcs_gemm_tn_graph", Wm.At, Wm.At.shape[1], feat, idx, B, C, N, k, Ctot, PD, ws, nbytes)
            if training and not _bn_fold(B * L):
                call("hpcs_vn_norm_stats", PD, B, Cout, Ctot, L, sums, 1)
                call("hpcs_bn_finalize", sums, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1)
            elif training:
                call("hpcs_vn_norm_stats_finalize", PD, B, Cout, Ctot, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd)
        elif k2:
            # two input channels (conv1): PD = W E is two fmas per value -- formed inside the point-wise kernels from the
            # 16 MB edge tensor, bit-identical to the GEMM's chain, instead of a 330 MB tensor written once and read 4 times
            PD, ldw = E, WxP.Wp.shape[1]
            if training and not _bn_fold(B * L):
                call("hpcs_vn_norm_stats_k2", E, WxP.Wp, ldw, B, Cout, Ctot, L, sums, 1)
                call("hpcs_bn_finalize", sums, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1)
            elif training:
                call("hpcs_vn_norm_stats_k2_finalize", E, WxP.Wp, ldw, B, Cout, Ctot, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean,
                     rvar, mean, rstd)
        else:
            PD = gemm_WX(WxP, E, B, 2 * C, 3 * L)
            del E
            if training and not _bn_fold(B * L):
                call("hpcs_vn_norm_stats", PD, B, Cout, Ctot, L, sums, 1)
                call("hpcs_bn_finalize", sums, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1)
            elif training:
                call("hpcs_vn_norm_stats_finalize", PD, B, Cout, Ctot, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd)
        if not training:
            call("hpcs_bn_finalize", None, Cout, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 0)
        out = torch.empty((B, Cout, 3, N) if pool_k else (B, Cout, 3, N, k), device=dev)
        if k2:
            call("hpcs_vn_apply_fwd_k2", PD, WxP.Wp, ldw, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, out)
        else:
            call("hpcs_vn_apply_fwd", PD, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, out)
        ctx.save_for_backward(feat, idx, WxP.Wp, PD, mean, rstd, gamma, beta)
        ctx.cfg = (B, C, N, k, Cout, Cd, pool_k, slope, flags, bool(training))
        ctx.k2 = k2
        ctx.tgraph = None
        if _SideWGrad.forward and graph_backward_supported(N, k) and any(ctx.needs_input_grad[i] for i in (0, 2, 3)):
            # the transposed graph of the backward's gradient reduction depends on idx only: built now, on the side stream
            nbytes = call("hpcs_edge_grad_reduce_workspace", B, Ctot, N)
            ws = torch.empty(nbytes // 4, device=dev)
            side = _SideWGrad.stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                call("hpcs_edge_graph_transpose", idx, B, Ctot, N, k, ws, nbytes)
                ev = torch.cuda.Event()
                ev.record()
            ctx.tgraph = (ws, nbytes, ev)
        return out

    @staticmethod
    def backward(ctx, g):
        feat, idx, Wp, PD, mean, rstd, gamma, beta = ctx.saved_tensors
        B, C, N, k, Cout, Cd, pool_k, slope, flags, training = ctx.cfg
        Ctot = Cout + Cd
        L = N * k
        g = _f32c(g)
        dev = g.device
        red = _zeros64(2 * Cout, dev)
        GPD = torch.empty(B * Ctot * 3 * L, device=dev)
        sg, sb = _bn_sinks(ctx.bnparams, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        if ctx.k2:                    # PD is the edge tensor E: pre-activations re-formed on the fly (see forward)
            ldw = Wp.shape[1]
            call("hpcs_vn_bwd_reduce_k2", PD, Wp, ldw, g, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, red, 1)
            call("hpcs_vn_bwd_apply_k2", PD, Wp, ldw, g, mean, rstd, gamma, beta, red, 1 if training else 0, B, Cout, Cd, L,
                 pool_k, slope, flags, GPD, sg, sb)
        else:
            call("hpcs_vn_bwd_reduce", PD, g, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, slope, flags, red, 1)
            call("hpcs_vn_bwd_apply", PD, g, mean, rstd, gamma, beta, red, 1 if training else 0, B, Cout, Cd, L, pool_k, slope,
                 flags, GPD, sg, sb)
        gUV = torch.empty(B, 2 * Ctot, 3, N, device=dev)
        if ctx.tgraph is not None:                                       # transposed graph built during the forward pass
            ws, nbytes, ev = ctx.tgraph
            torch.cuda.current_stream().wait_event(ev)
            call("hpcs_edge_grad_reduce_prebuilt", GPD, B, Ctot, N, k, gUV, ws, nbytes)
            ctx.tgraph = None
        else:
            nbytes = call("hpcs_edge_grad_reduce_workspace", B, Ctot, N)
            ws = torch.empty(nbytes // 4, device=dev)
            call("hpcs_edge_grad_reduce", GPD, idx, B, Ctot, N, k, gUV, ws, nbytes)
        del GPD
        gfeat = gWf = gWd = None
        W = Wp[:Ctot, :2 * C]                                            # [Wf;Wd] rows, (difference | centre) columns
        if ctx.needs_input_grad[0]:
            Wuv = torch.empty(2 * Ctot, C, device=dev)                               # [Wa ; Wb - Wa]
            call("hpcs_graph_wuv", Wp, Wp.shape[1], Ctot, C, Wuv)
            gfeat = gemm_WtG(Wuv, gUV, B, 2 * Ctot, 3 * N).view_as(feat)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            def weight_grads():
                gWuv = gemm_GXt(gUV, feat, B, 2 * Ctot, C, 3 * N)                    # [2Ctot, C]
                return torch.cat([gWuv[:Ctot] - gWuv[Ctot:], gWuv[Ctot:]], dim=1)     # [Ctot, 2C]

            Wf_, Wd_ = ctx.wparams
            sink_f = _grad_sink((Wf_, ...)) if ctx.needs_input_grad[2] else None
            sink_d = _grad_sink((Wd_, ...)) if ctx.needs_input_grad[3] else None
            if (sink_f is not None or not ctx.needs_input_grad[2]) and (sink_d is not None or not ctx.needs_input_grad[3]) and \
                    all(t is None or t.is_contiguous() for t in (sink_f, sink_d)):
                def work():
                    gWuv = gemm_GXt(gUV, feat, B, 2 * Ctot, C, 3 * N)                # [2Ctot, C]
                    call("hpcs_graph_wgrad_fold", gWuv, Ctot, C, Cout, sink_f, sink_d)    # [gU - gV | gV] added into .grad
                _on_side_stream(work, (gUV, feat))
            else:
                gWcat = weight_grads()
                gWf, gWd = gWcat[:Cout], gWcat[Cout:]
        ggamma, gbeta = (None, None) if sg is not None else \
            _bn_grads(ctx.bnparams, red, Cout, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        return gfeat, None, gWf, gWd, ggamma, gbeta, None, None, None, None, None


def graph_backward_supported(N, k):
    """Geometries for which hpcs_edge_grad_reduce runs over the transposed graph (csrc/vn_layers.hip: one N*k plane, or one
    slab of it, of 20480 floats in LDS); elsewhere it falls back to atomics and the edge-level backward is faster."""
    L = N * k
    if k % 4:
        return False
    if N <= 1024 and L <= 20480:
        return True
    if N <= 2048 and L % 20480 == 0:
        S = L // 20480
        return S <= 8 and N % S == 0 and (N // S) * k == 20480
    return False


def vn_graph_layer(feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope=0.2, pool_k=0):
    _need_cuda(feat, idx, Wf)
    return VNGraphLayerFn.apply(feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool_k)


def vn_edge_layer(feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope=0.2, pool=False):
    _need_cuda(feat, idx, Wf)
    return VNEdgeLayerFn.apply(feat, idx, Wf, Wd, gamma, beta, rmean, rvar, training, slope, pool)


# ------------------------------------------------------------------------------------------------ Conv1d heads
class BnActFn(torch.autograd.Function):
    """BatchNorm1d over [B,C,L] (+ LeakyReLU(slope)) (* mask)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, rmean, rvar, training, slope, mask, drop=None):
        """mask: explicit [B,C,L] multiplier; drop = (seed, p[, seed_dev]): the multiplier is generated inside the
        kernels; seed_dev (int64[1] device tensor) overrides the by-value seed (graph-replayable launches)."""
        ctx.bnparams = (gamma, beta)
        dseed, dp = (int(drop[0]) & (2 ** 63 - 1), float(drop[1])) if (drop is not None and mask is None) else (0, 0.0)
        dsd = drop[2] if (drop is not None and mask is None and len(drop) > 2) else None
        ctx.drop = (dseed, dp, dsd)
        y = _f32c(y)
        B, C = y.shape[0], y.shape[1]
        L = y[0, 0].numel()
        dev = y.device
        mean = torch.empty(C, device=dev)
        rstd = torch.empty(C, device=dev)
        if training and not _bn_fold(B * L):
            sums = _zeros64(2 * C, dev)
            call("hpcs_bn_stats", y, B, C, L, sums, 1)
            call("hpcs_bn_finalize", sums, C, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 1)
        elif training:                     # statistics + finalisation in one launch (the last workgroup finalises)
            sums, ticket = _zeros64_ticket(2 * C, dev)
            call("hpcs_bn_stats_finalize", y, B, C, L, sums, ticket, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd)
        else:
            call("hpcs_bn_finalize", None, C, B * L, BN_EPS, BN_MOMENTUM, rmean, rvar, mean, rstd, 0)
        out = torch.empty_like(y)
        call("hpcs_bn_act_fwd", y, mean, rstd, gamma, beta, mask, dseed, dsd, dp, B, C, L, slope, out)
        ctx.save_for_backward(y, mean, rstd, gamma, beta, mask)
        ctx.cfg = (B, C, L, slope, bool(training))
        return out

    @staticmethod
    def backward(ctx, g):
        y, mean, rstd, gamma, beta, mask = ctx.saved_tensors
        B, C, L, slope, training = ctx.cfg
        g = _f32c(g)
        red = _zeros64(2 * C, g.device)
        dseed, dp, dsd = ctx.drop
        call("hpcs_bn_act_bwd_reduce", y, g, mean, rstd, gamma, beta, mask, dseed, dsd, dp, B, C, L, slope, red, 1)
        gy = torch.empty_like(y)
        sg, sb = _bn_sinks(ctx.bnparams, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        call("hpcs_bn_act_bwd_apply", y, g, mean, rstd, gamma, beta, mask, dseed, dsd, dp, red, 1 if training else 0, B, C, L,
             slope, gy, sg, sb)
        ggamma, gbeta = (None, None) if sg is not None else \
            _bn_grads(ctx.bnparams, red, C, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gy, ggamma, gbeta, None, None, None, None, None, None


def bn_act(y, gamma, beta, rmean, rvar, training, slope=0.2, mask=None, drop=None):
    _need_cuda(y)
    return BnActFn.apply(y, gamma, beta, rmean, rvar, training, slope, mask, drop)


def dropout_mask(seed, p, shape, device):
    """The multiplier tensor that bn_act(..., drop=(seed, p)) applies implicitly (tests)."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("hpcs_dropout_mask", int(seed) & (2 ** 63 - 1), float(p), out.numel(), out)
    return out


# ------------------------------------------------------------------------------------------------ glue ops
class ConcatMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        B, C, _, N = x.shape
        out = torch.empty(B, 2 * C, 3, N, device=x.device)
        call("hpcs_concat_mean_fwd", x, B, C, N, out)
        ctx.dims = (B, C, N)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, N = ctx.dims
        gx = torch.empty(B, C, 3, N, device=g.device)
        call("hpcs_concat_mean_bwd", _f32c(g), B, C, N, gx)
        return gx


class FrameProjectFn(torch.autograd.Function):
    """einsum('bijm,bjkm->bikm', x [B,C,3,N], z [B,3,3,N])."""

    @staticmethod
    def forward(ctx, x, z):
        x, z = _f32c(x), _f32c(z)
        B, C, _, N = x.shape
        out = torch.empty_like(x)
        call("hpcs_frame_project_fwd", x, z, B, C, N, out)
        ctx.save_for_backward(x, z)
        return out

    @staticmethod
    def backward(ctx, g):
        x, z = ctx.saved_tensors
        B, C, _, N = x.shape
        gx = torch.empty_like(x)
        gz = torch.empty_like(z)
        call("hpcs_frame_project_bwd", x, z, _f32c(g), B, C, N, gx, gz)
        return gx, gz


class ProjectMaxFn(torch.autograd.Function):
    """max over N of einsum('bijm,bjkm->bikm', cat(x, xconst expanded over N), z) -> [B, (C+Cc)*3], fused."""

    @staticmethod
    def forward(ctx, x, xconst, z, hub=None):
        ctx.hub = hub
        x, z = _f32c(x), _f32c(z)
        B, C, _, N = x.shape
        Cc = 0 if xconst is None else xconst.shape[1]
        xc = None if xconst is None else _f32c(xconst)
        out = torch.empty(B, (C + Cc) * 3, device=x.device)
        arg = torch.empty(B, (C + Cc) * 3, device=x.device, dtype=torch.int32)
        call("hpcs_project_max_fwd", x, xc, z, B, C, Cc, N, out, arg)
        ctx.save_for_backward(x, xc, z, arg)
        return out

    @staticmethod
    def backward(ctx, g):
        x, xc, z, arg = ctx.saved_tensors
        B, C, _, N = x.shape
        Cc = 0 if xc is None else xc.shape[1]
        g = _f32c(g)
        gz = torch.empty_like(z)
        gxc = None if xc is None else torch.empty_like(xc)
        if ctx.hub is not None and ctx.needs_input_grad[0]:
            # x part: queued for the producer of x's dense gradient (GradHub); z / xconst parts now
            call("hpcs_project_max_bwd", x, xc, z, arg, g, B, C, Cc, N, None, 0, gxc, gz)
            ctx.hub.give(lambda gx: call("hpcs_project_max_bwd", x, xc, z, arg, g, B, C, Cc, N, gx, 1, None, None))
            return None, gxc, gz, None
        gx = torch.empty_like(x)
        call("hpcs_project_max_bwd", x, xc, z, arg, g, B, C, Cc, N, gx, 0, gxc, gz)
        return gx, gxc, gz, None


class RowMaxFn(torch.autograd.Function):
    """max over the last dim (values only differentiable)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        L = x.shape[-1]
        rows = x.numel() // L
        out = torch.empty(x.shape[:-1], device=x.device)
        arg = torch.empty(rows, device=x.device, dtype=torch.int32)
        call("hpcs_rowmax_fwd", x, rows, L, out, arg)
        ctx.save_for_backward(arg)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        L = ctx.shape[-1]
        rows = arg.numel()
        gx = torch.empty(ctx.shape, device=g.device)
        call("hpcs_rowmax_bwd", _f32c(g), arg, rows, L, gx)
        return gx


class RowMeanFn(torch.autograd.Function):
    """mean over the last dim."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        L = x.shape[-1]
        rows = x.numel() // L
        out = torch.empty(x.shape[:-1], device=x.device)
        call("hpcs_rowsum", x, rows, L, 1.0 / L, out)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        L = ctx.shape[-1]
        return (g / L).unsqueeze(-1).expand(ctx.shape)


class RowSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        L = x.shape[-1]
        rows = x.numel() // L
        out = torch.empty(x.shape[:-1], device=x.device)
        call("hpcs_rowsum", x, rows, L, 1.0, out)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(-1).expand(ctx.shape)


class VNMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, d):
        x, d = _f32c(x), _f32c(d)
        B, C, _, N, k = x.shape
        out = torch.empty(B, C, 3, N, device=x.device)
        arg = torch.empty(B, C, N, device=x.device, dtype=torch.int32)
        call("hpcs_vn_maxpool_fwd", x, d, B, C, N, k, out, arg)
        ctx.save_for_backward(arg)
        ctx.dims = (B, C, N, k)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, C, N, k = ctx.dims
        gx = torch.empty(B, C, 3, N, k, device=g.device)
        call("hpcs_vn_maxpool_bwd", _f32c(g), arg, B, C, N, k, gx)
        return gx, None
