#!/usr/bin/env python3
"""A/B two builds of the library on one train-mode extractor forward + backward: bitwise comparison of the taps.
usage: ab_lib.py run <lib.so> <out.pt>   |   ab_lib.py cmp <a.pt> <b.pt>
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(lib, out):
    import hpcs_amd._lib as L
    L.LIB_PATH = os.path.abspath(lib)
    from _det import det_cloud, det_int
    from hpcs_amd.nn.dgcnn import VN_DGCNN_partseg
    torch.manual_seed(3)
    net = VN_DGCNN_partseg(in_channels=3, out_features=32, k=10, dropout=0.0, pooling="mean", num_categories=16).cuda().train()
    x = det_cloud(4, 256, 7100).cuda()
    lab = torch.eye(16)[det_int((4,), 7200, 16)].cuda()
    y = net(x, lab)
    y.square().sum().backward()
    res = {"y": y.detach().cpu()}
    for n, p in net.named_parameters():
        if p.grad is not None:
            res["g_" + n] = p.grad.detach().cpu()
    torch.save(res, out)


def layer(lib, out):
    import hpcs_amd._lib as L
    L.LIB_PATH = os.path.abspath(lib)
    from hpcs_amd.nn.dgcnn.utils.vn_layers import VNLinearLeakyReLU
    torch.manual_seed(5)
    m = VNLinearLeakyReLU(21, 21, dim=5).cuda().train()
    x = torch.randn(4, 21, 3, 256, 10, device="cuda", requires_grad=True)
    y = m(x)
    (y * torch.randn_like(y)).sum().backward()
    m.eval()
    with torch.no_grad():
        ye = m(x)
    res = {"y_train": y.detach().cpu(), "y_eval": ye.cpu(), "gx": x.grad.cpu()}
    for n, p in m.named_parameters():
        res["g_" + n] = p.grad.detach().cpu()
    torch.save(res, out)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in A:
        d = (A[k] != B[k]).sum().item()
        rel = ((A[k] - B[k]).norm() / (A[k].norm() + 1e-30)).item()
        print(f"{k:40s} differing elements {d:8d} of {A[k].numel():8d}  rel L2 {rel:.2e}")


if __name__ == "__main__":
    {"run": run, "layer": layer, "cmp": cmp}[sys.argv[1]](*sys.argv[2:])
