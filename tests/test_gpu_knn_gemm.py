"""GPU parity: kNN (bit-exact vs the oracle) and the MFMA GEMMs, called through the C ABI."""
import os

import numpy as np
import pytest
import torch

from _det import det_cloud, det_normalish, det_uniform
from hpcs_amd import _lib
from oracle import extractor as OE

pytestmark = pytest.mark.gpu
dev = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hip_knn(x, k):
    B, D, N = x.shape
    idx = torch.empty(B, N, k, dtype=torch.int64, device=dev)
    nbytes = _lib.call("hpcs_knn_workspace", B, N)
    ws = torch.empty(nbytes // 4, device=dev)
    _lib.call("hpcs_knn", x.to(dev).contiguous(), B, D, N, k, idx, ws, nbytes)
    idx2 = torch.empty_like(idx)                      # scalar fallback kernel (no workspace) must agree
    _lib.call("hpcs_knn", x.to(dev).contiguous(), B, D, N, k, idx2, None, 0)
    assert torch.equal(idx, idx2)
    return idx.cpu()


@pytest.mark.parametrize("B,D,N,k,seed", [(2, 3, 256, 10, 11), (4, 3, 1024, 20, 14), (2, 63, 256, 10, 12),
                                          (2, 63, 1024, 20, 15), (1, 3, 100, 7, 16), (2, 6, 77, 5, 17),
                                          (1, 3, 2048, 40, 18), (1, 63, 2048, 40, 19), (3, 3, 64, 64, 20)])
def test_knn_bit_exact(B, D, N, k, seed):
    x = det_cloud(B, N, seed) if D == 3 else det_normalish((B, D, N), seed)
    want = OE.knn_exact(x, k)
    got = hip_knn(x, k)
    assert torch.equal(got, want)


@pytest.mark.parametrize("kernel", ["tiled", "queue", "bq", "bq2"])
def test_knn_both_mfma_kernels_bit_exact(kernel, tuning):
    """The MFMA selection kernels (interleaved 4-wave tiles with a sorting network / one wave with
    per-lane queues / one wave with a stack of pending candidates merged 16 at a time) are chosen by problem size and k;
    each must reproduce the oracle on every shape, duplicates included ("bq" serves k <= 32: larger k falls through to the
    queued kernel; "bq2" = two waves per query tile, and for k = 33..40 32-wide lists with the flagged repair pass)."""
    tuning("HPCS_KNN_KERNEL", _lib.KNN_KERNEL_CODES[kernel])
    for B, D, N, k, seed in [(2, 3, 256, 10, 31), (3, 3, 1024, 20, 32), (2, 63, 512, 20, 33), (1, 63, 2048, 40, 34),
                             (2, 3, 100, 7, 35), (1, 3, 64, 33, 36)]:
        x = det_cloud(B, N, seed) if D == 3 else det_normalish((B, D, N), seed)
        assert torch.equal(hip_knn(x, k), OE.knn_exact(x, k)), (kernel, B, D, N, k)
    x = det_cloud(2, 128, 37, dup_every=5)
    assert torch.equal(hip_knn(x, 8), OE.knn_exact(x, 8))
    # ties everywhere (a lattice: many equal distances), k = 1 / 4 / 32, a ragged last tile and a query tile beyond N
    g = torch.arange(5, dtype=torch.float32)
    lat = torch.stack(torch.meshgrid(g, g, g, indexing="ij")).reshape(1, 3, 125)
    for k_ in (1, 4, 32):
        assert torch.equal(hip_knn(lat, k_), OE.knn_exact(lat, k_)), (kernel, "lattice", k_)
    x = det_cloud(3, 45, 38)
    assert torch.equal(hip_knn(x, 16), OE.knn_exact(x, 16))


def test_knn_batch_queue_kernel_is_the_default_for_many_queries():
    """B * N / 32 >= 1024 query tiles and k <= 32 select the batch-queue kernel without any switch: same indices as the
    oracle (and so as the sorting-network kernel, which the small shapes above take)."""
    x = det_cloud(36, 1024, 39)
    assert torch.equal(hip_knn(x, 20), OE.knn_exact(x, 20))


@pytest.mark.parametrize("D", [3, 63])
def test_knn_k40_repair_pass_when_one_sublist_holds_all_neighbours(D):
    """k in 33..40 runs the sorting-network kernel on 32-wide per-lane lists and merges a query's k nearest from 8
    sub-lists (4 waves x 2 lane halves, interleaved candidate tiles); that is exact unless ONE sub-list would have to supply
    more than 32 of them -- then the merge exhausts it, the tile is flagged and the queued kernel recomputes it.  Adversarial
    cloud: the 40 mutually nearest points all sit at candidate indices of sub-list (wave 0, half 0), i.e. tile % 4 == 0 and
    (position in tile / 4) % 2 == 0, so the repair pass MUST run for their tiles; every query is still checked bit-exactly."""
    N, k = 2048, 40
    x = det_cloud(1, N, 71) if D == 3 else det_normalish((1, D, N), 72)
    special = [128 * a + p for a in range(3) for p in (0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27)][:40]
    tight = det_uniform((D, 40), 73) * 1e-3               # a tight cluster far from the rest of the cloud
    x[0][:, special] = tight + 5.0
    want = OE.knn_exact(x, k)
    got = hip_knn(x, k)
    assert torch.equal(got, want)
    assert set(got[0, special[0]].tolist()) == set(special)          # the cluster's neighbours are the cluster


def test_knn_k40_neighbours_split_over_both_lists_of_one_wave():
    """k = 33..40 at D=3 runs the two-wave batch-queue kernel on 32-wide lists.  Its cross-list admission bound (a key with 16
    better keys in each of the wave's two lists is out) is exact only for k <= 32: for k = 40 such a key can still rank
    33..40 and neither list runs dry, so the repair pass never fires.  Adversarial cloud on a line: query 0's 40
    nearest are 17 points in (wave 0, lane half 0), 17 in (wave 0, half 1), 5 in wave 1 -- and ONE point that arrives at a
    late candidate tile of wave 0 with 17 + 17 better keys already listed and must still be returned (rank 38)."""
    N, k = 2048, 40
    xs = 1.0 + torch.arange(N, dtype=torch.float32) * 1e-3                 # everything far from the query by default
    half0 = [p for p in range(32) if p % 8 < 4]
    half1 = [p for p in range(32) if p % 8 >= 4]
    near0 = half0 + [64]                                                   # tiles 0 and 2: wave 0 (even tiles), half 0
    near1 = half1 + [68]
    rank = 0
    for a, b in zip(near0, near1):                                         # 34 near points, alternating between the lists
        xs[a] = 1e-3 * rank
        xs[b] = 1e-3 * (rank + 1)
        rank += 2
    wave1 = [32 + 1, 32 + 5, 96 + 2, 96 + 6, 160 + 3]                     # odd tiles: wave 1
    for i, j in enumerate(wave1[:3]):
        xs[j] = 0.040 + 1e-3 * i                                           # ranks 35..37
    late = 60 * 32                                                         # tile 60 (wave 0), position 0 (half 0)
    xs[late] = 0.045                                                       # rank 38
    for i, j in enumerate(wave1[3:]):
        xs[j] = 0.046 + 1e-3 * i                                           # ranks 39, 40
    assert xs[0] == 0.0
    x = torch.zeros(1, 3, N)
    x[0, 0] = xs
    want = OE.knn_exact(x, k)
    assert late in want[0, 0].tolist() and want[0, 0].tolist().index(late) == 37
    got = hip_knn(x, k)
    assert late in got[0, 0].tolist()
    assert torch.equal(got, want)


def test_knn_duplicates_and_golden(golden):
    x = det_cloud(2, 128, 13, dup_every=5)
    assert torch.equal(hip_knn(x, 8), OE.knn_exact(x, 8))
    x = det_cloud(2, 256, 11)
    assert np.array_equal(hip_knn(x, 10).numpy(), golden["knn_d3_idx"].astype(np.int64))
    x = det_normalish((2, 63, 256), 12)
    assert np.array_equal(hip_knn(x, 10).numpy(), golden["knn_d63_idx"].astype(np.int64))


def test_knn_errors():
    x = torch.zeros(1, 3, 16, device=dev)
    idx = torch.empty(1, 16, 20, dtype=torch.int64, device=dev)
    with pytest.raises(_lib.HpcsNativeError):
        _lib.call("hpcs_knn", x, 1, 3, 16, 20, idx, None, 0)       # k > N
    with pytest.raises(_lib.HpcsNativeError):
        _lib.call("hpcs_knn", torch.zeros(1, 3, 16), 1, 3, 16, 4, idx, None, 0)   # CPU tensor: no fallback


# shapes with < 32 MFMA tiles take the per-element kernel, the larger ones the 64x128 / 128x128 MFMA kernels
@pytest.mark.parametrize("M,K,Nc,nb", [(21, 2, 300, 3), (42, 42, 1024, 2), (341, 63, 3 * 256, 2),
                                       (682, 682, 3 * 128, 2), (256, 189, 1024, 2), (3, 170, 96, 4), (64, 16, 1, 5),
                                       (42, 42, 2560, 2), (21, 2, 2600, 3), (341, 63, 1536, 4), (682, 682, 768, 2),
                                       (256, 189, 2051, 2), (96, 341, 341, 1)])
def test_gemm_tn(M, K, Nc, nb):
    W = det_uniform((M, K), 200 + M)           # weights [M,K]
    X = det_normalish((nb, K, Nc), 300 + K)
    bias = det_uniform((nb, M), 400)
    At = W.t().contiguous().to(dev)
    C = torch.empty(nb, M, Nc, device=dev)
    _lib.call("hpcs_gemm_tn", At, M, X.to(dev), Nc, K * Nc, 0, C, Nc, M * Nc, 0, M, Nc, K, nb, 1, bias.to(dev), 0)
    want = torch.matmul(W.double(), X.double()) + bias.double().unsqueeze(-1)
    torch.testing.assert_close(C.cpu().double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    # data-grad form: At = W, reduces over M
    G = det_normalish((nb, M, Nc), 500)
    gX = torch.full((nb, K, Nc), 7.0, device=dev)
    _lib.call("hpcs_gemm_tn", W.to(dev), K, G.to(dev), Nc, M * Nc, 0, gX, Nc, K * Nc, 0, K, Nc, M, nb, 1, None, 1)
    want = torch.matmul(W.t().double(), G.double()) + 7.0
    torch.testing.assert_close(gX.cpu().double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_gemm_tn_inner_batch():
    # walk a [B,K,3,N] tensor as 3B matrices of N columns with a per-(b,xyz) row bias
    B, K, N, M = 2, 10, 96, 7
    W = det_uniform((M, K), 1)
    X = det_normalish((B, K, 3, N), 2)
    bias = det_uniform((B * 3, M), 3)
    C = torch.empty(B, M, 3, N, device=dev)
    _lib.call("hpcs_gemm_tn", W.t().contiguous().to(dev), M, X.to(dev), 3 * N, K * 3 * N, N, C, 3 * N, M * 3 * N, N,
              M, N, K, B, 3, bias.to(dev), 0)
    want = torch.einsum("mk,bkxn->bmxn", W.double(), X.double()) + bias.double().view(B, 3, M).permute(0, 2, 1).unsqueeze(-1)
    torch.testing.assert_close(C.cpu().double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,Kx,Nc,nb", [(42, 42, 5000, 3), (341, 63, 3 * 256, 2), (682, 682, 3 * 64, 2), (256, 1, 1, 6),
                                        (3, 170, 96, 4),
                                        # gemm_nt_big_kernel (outputs from 300 x 300 up): VNStdFeature's two shapes with ragged tile
                                        # edges (682 = 5 x 128 + 42, 341 = 2 x 128 + 85), a column count that is not a multiple
                                        # of the 16-deep step, odd (unaligned) rows, fewer steps than splits
                                        (682, 341, 3072, 3), (340, 341, 1000, 3), (301, 303, 77, 2), (300, 300, 8, 1)])
def test_gemm_nt(M, Kx, Nc, nb):
    G = det_normalish((nb, M, Nc), 600 + M)
    X = det_normalish((nb, Kx, Nc), 700 + Kx)
    nbytes = _lib.call("hpcs_gemm_nt_workspace", M, Kx, Nc, nb, 1)
    ws = torch.empty(nbytes // 4 + 1, device=dev)
    dW = torch.full((M, Kx), 3.0, device=dev)
    _lib.call("hpcs_gemm_nt", G.to(dev), Nc, M * Nc, 0, X.to(dev), Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, dW, Kx, 1, ws, nbytes)
    want = torch.einsum("bmc,bkc->mk", G.double(), X.double()) + 3.0
    torch.testing.assert_close(dW.cpu().double(), want, rtol=1e-5, atol=2e-6 * float(want.abs().max()) * np.sqrt(Nc * nb))


@pytest.mark.parametrize("M,Kx,Nc,nb", [(42, 42, 5000, 3), (42, 21, 61440, 8), (341, 63, 3 * 256, 2), (682, 682, 3 * 64, 2),
                                        (256, 1, 1, 6), (3, 170, 96, 4), (64, 256, 1024, 32), (340, 341, 1536, 2)])
def test_gemm_nt_fused_reduction_equals_the_two_launch_form(M, Kx, Nc, nb):
    """hpcs_gemm_nt_fused: the split partials are summed by the last-arriving workgroup inside the launch (per slice of
    splits, then per tile; write-through partials, agent-scope tickets).  Checked against fp64, against the separate
    split_reduce launches (same partials, another fixed order: equal to fp32 rounding of the sum), bit-reproducible run to
    run, with the SAME ticket words reused by a second call (the kernel leaves them zero) and accumulate on / off.
    (42, 21, 61440, 8) is the EdgeConv weight-gradient shape: one tile, hundreds of splits, two reduction levels."""
    G = det_normalish((nb, M, Nc), 600 + M).to(dev)
    X = det_normalish((nb, Kx, Nc), 700 + Kx).to(dev)
    nbytes = _lib.call("hpcs_gemm_nt_workspace", M, Kx, Nc, nb, 1)
    nt = _lib.call("hpcs_gemm_nt_tickets", M, Kx, Nc, nb, 1)
    assert nt >= 1
    ws = torch.empty(nbytes // 4 + 1, device=dev)
    tickets = torch.zeros(nt, dtype=torch.int32, device=dev)
    want = torch.einsum("bmc,bkc->mk", G.double().cpu(), X.double().cpu())
    tol = dict(rtol=1e-5, atol=2e-6 * float(want.abs().max()) * np.sqrt(Nc * nb))
    outs = []
    for acc in (0, 1, 1):
        dW = torch.full((M, Kx), 3.0, device=dev)
        _lib.call("hpcs_gemm_nt_fused", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, dW, Kx, acc, ws, nbytes, tickets, nt)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0                               # left zero for the next call
        torch.testing.assert_close(dW.cpu().double(), want + (3.0 if acc else 0.0), **tol)
        outs.append(dW.clone())
    assert torch.equal(outs[1], outs[2])                                   # fixed summation order
    ref = torch.full((M, Kx), 3.0, device=dev)
    _lib.call("hpcs_gemm_nt", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, ref, Kx, 1, ws, nbytes)
    torch.testing.assert_close(outs[1], ref, rtol=2e-6, atol=1e-6 * float(want.abs().max()))
    with pytest.raises(_lib.HpcsNativeError):
        _lib.call("hpcs_gemm_nt_fused", G, Nc, M * Nc, 0, X, Nc, Kx * Nc, 0, M, Kx, Nc, nb, 1, ref, Kx, 1, ws, nbytes, tickets, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("R,K,M,ld", [(32, 2110, 256, 2299), (96, 341, 682, 682), (5, 7, 3, 9)])
def test_row_linear_matches_fp64(R, K, M, ld):
    """ops.row_linear: out = vec @ W^T on a column slice of a wider weight, forward and both gradients."""
    from hpcs_amd import ops
    torch.manual_seed(3)
    Wfull = torch.randn(M, ld, device="cuda", requires_grad=True)
    vec = torch.randn(R, K, device="cuda", requires_grad=True)
    out = ops.row_linear(vec, Wfull[:, ld - K:])
    g = torch.randn_like(out)
    out.backward(g)
    ref = vec.detach().double() @ Wfull.detach()[:, ld - K:].double().t()
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    gv = g.double() @ Wfull.detach()[:, ld - K:].double()
    gw = g.double().t() @ vec.detach().double()
    assert torch.allclose(vec.grad.double(), gv, rtol=1e-5, atol=1e-5 * float(gv.abs().max()))
    assert torch.allclose(Wfull.grad[:, ld - K:].double(), gw, rtol=1e-5, atol=1e-5 * float(gw.abs().max()))
    assert float(Wfull.grad[:, :ld - K].abs().max()) == 0.0 if ld > K else True


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,N,k,M", [(3, 21, 256, 20, 42), (2, 5, 96, 8, 10), (1, 21, 1024, 20, 42), (4, 32, 128, 4, 64), (3, 7, 200, 12, 3)])
def test_gemm_tn_graph_equals_gather_then_gemm_bit_for_bit(B, C, N, k, M):
    """hpcs_gemm_tn_graph (edge tensor formed in the operand loader) against hpcs_edge_gather_fwd + hpcs_gemm_tn: the same
    values through the same k-ordered fma chain, so the outputs must be IDENTICAL (get_graph_feature, vn_dgcnn_util.py:10-37).
    (Shapes with >= 32 tiles of 128 x 128: below that hpcs_gemm_tn takes its per-element kernel, another summation order.)"""
    from hpcs_amd._lib import call
    torch.manual_seed(B * 1000 + C)
    dev = "cuda"
    feat = torch.randn(B, C, 3, N, device=dev)
    idx = torch.randint(0, N, (B, N, k), device=dev, dtype=torch.int64)
    W = torch.randn(M, 2 * C, device=dev)
    lda = (M + 3) // 4 * 4
    At = torch.zeros(2 * C, lda, device=dev)
    At[:, :M] = W.t()
    L = N * k
    E = torch.empty(B, 2 * C, 3, N, k, device=dev)
    call("hpcs_edge_gather_fwd", feat, idx, B, C, N, k, E)
    ref = torch.empty(B, M, 3 * L, device=dev)
    call("hpcs_gemm_tn", At, lda, E, 3 * L, 2 * C * 3 * L, 0, ref, 3 * L, M * 3 * L, 0, M, 3 * L, 2 * C, B, 1, None, 0)
    got = torch.full((B, M, 3 * L), float("nan"), device=dev)
    nbytes = call("hpcs_gemm_tn_graph_workspace", B, C, N)
    ws = torch.empty(nbytes // 4, device=dev)
    call("hpcs_gemm_tn_graph", At, lda, feat, idx, B, C, N, k, M, got, ws, nbytes)
    assert -(-M // 128) * -(-3 * L // 128) * B >= 32
    assert torch.equal(got, ref)
    # and against the definition in fp64
    want = torch.einsum("mk,bkl->bml", W.double(), E.reshape(B, 2 * C, 3 * L).double())
    assert (got.double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.gpu
def test_gemm_tn_graph_refuses_what_its_loader_cannot_do():
    from hpcs_amd._lib import call, HpcsNativeError
    dev = "cuda"
    feat = torch.randn(1, 4, 3, 32, device=dev)
    At = torch.zeros(8, 68, device=dev)
    out = torch.empty(1, 65, 3 * 32 * 8, device=dev)
    ws = torch.empty(1024, device=dev)
    with pytest.raises(HpcsNativeError):                                   # k % 4 != 0
        call("hpcs_gemm_tn_graph", At, 68, feat, torch.zeros(1, 32, 6, dtype=torch.int64, device=dev), 1, 4, 32, 6, 8, out, ws, 4096)
    with pytest.raises(HpcsNativeError):                                   # more than one 64-row tile
        call("hpcs_gemm_tn_graph", At, 68, feat, torch.zeros(1, 32, 8, dtype=torch.int64, device=dev), 1, 4, 32, 8, 65, out, ws, 4096)
