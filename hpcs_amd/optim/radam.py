"""Mirror of hpcs/optim/radam.py:29-134 (Riemannian Adam applied to every parameter) as ONE fused HIP launch.

Quirks of the reference that are kept: group["step"] is incremented twice per parameter
(radam.py:116,133), so parameter i of a step uses the exponent 2*n_done+1; project() clips every row of every
tensor to norm 1-4e-3; weight decay and egrad2rgrad mutate .grad in place.
"""
import ctypes

import torch

from .. import _lib


class RAdam(torch.optim.Adam):
    def _tables(self, group, params):
        """Device tables (descriptors + row starts), rebuilt every step because the step sizes change.  Tensors whose
        points are one-dimensional (Conv1d weights [Co,Ci,1], `scale`) go to a table walked with one lane per row,
        rows of up to 64 columns with eight lanes per row, wider rows with a whole wave per row; the step sizes follow the reference's parameter order either way."""
        import struct
        b1, b2 = group["betas"]
        lr = group["lr"]
        entries = {1: [], 8: [], 64: []}
        for p in params:
            st = self.state[p]
            group["step"] += 1
            t = group["step"]
            step_size = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
            group["step"] += 1
            cols = p.shape[-1] if p.dim() > 0 else 1
            rows = p.numel() // cols
            entries[1 if cols == 1 else (8 if cols <= 64 else 64)].append(
                (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), rows, cols, step_size))
        out = []
        for lanes, ents in entries.items():
            if not ents:
                continue
            blob = bytearray()
            row_starts, rows_total = [], 0
            for e in ents:
                blob += struct.pack("<QQQQqif", *e)
                row_starts.append(rows_total)
                rows_total += e[4]
            row_starts.append(rows_total)
            blob += struct.pack(f"<{len(row_starts)}q", *row_starts)
            out.append((lanes, torch.frombuffer(blob, dtype=torch.uint8), len(ents), rows_total))
        return out

    def _live_params(self, group):
        params = []
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise RuntimeError("Riemannian Adam does not support sparse gradients")
            if not p.is_cuda:
                raise _lib.HpcsNativeError("hpcs_amd.optim.RAdam needs HIP-device parameters (no CPU fallback)")
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise _lib.HpcsNativeError("RAdam needs contiguous parameters and gradients")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            params.append(p)
        return params

    def build_tables(self):
        """Host half of a step, callable BEFORE the gradients exist (the tables hold addresses and step sizes only):
        advances group["step"] like the reference and returns [(lanes, host blob, n_tensors, rows_total)] for the single
        parameter group HPCS uses (base_hyp_hc.py:143).  Pair with step(tables_dev=...)."""
        if len(self.param_groups) != 1:
            raise NotImplementedError("build_tables() serves the single parameter group HPCS configures")
        group = self.param_groups[0]
        group.setdefault("step", 0)
        if group.get("amsgrad", False):
            raise NotImplementedError("amsgrad is never enabled by HPCS (base_hyp_hc.py:143)")
        return self._tables(group, self._live_params(group))

    @torch.no_grad()
    def step(self, closure=None, tables=None, tables_dev=None):
        """tables / tables_dev: what build_tables() returned and the persistent device copies of its blobs (the step
        driver's parameter block); without them the tables are built and uploaded here."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if tables is not None:
            group = self.param_groups[0]
            b1, b2 = group["betas"]
            for (lanes, _, n_tensors, rows_total), dev_table in zip(tables, tables_dev):
                _lib.call("hpcs_radam_step", dev_table, n_tensors, rows_total, lanes, b1, b2, group["eps"],
                          group["weight_decay"])
            return loss
        for group in self.param_groups:
            if "step" not in group:
                group["step"] = 0
            if group.get("amsgrad", False):
                raise NotImplementedError("amsgrad is never enabled by HPCS (base_hyp_hc.py:143)")
            params = self._live_params(group)
            if not params:
                continue
            from .._pinned import upload
            b1, b2 = group["betas"]
            keep = []
            for lanes, host, n_tensors, rows_total in self._tables(group, params):
                dev_table = upload(host, params[0].device, key=f"radam{lanes}")       # async, through a pinned ring
                _lib.call("hpcs_radam_step", dev_table, n_tensors, rows_total, lanes, b1, b2, group["eps"],
                          group["weight_decay"])
                keep.append(dev_table)
            self._keep = keep          # keep the tables alive until the kernels have consumed them
        return loss
