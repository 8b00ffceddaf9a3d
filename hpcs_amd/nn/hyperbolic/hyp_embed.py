"""Mirror of hpcs/nn/hyperbolic/hyp_embed.py:6-23 on the fused exp-map kernel."""
import torch.nn as nn

from .. import MLP
from ...utils.poincare import expmap0


class ExpMap(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x):
        return expmap0(x)


class MLPExpMap(nn.Module):
    def __init__(self, input_feat: int, out_feat: int, bias: bool = False, negative_slope: float = 0.2,
                 dropout: float = 0.0):
        super().__init__()
        self.mlp = MLP([input_feat, out_feat], bias=bias, negative_slope=negative_slope, dropout=dropout)

    def forward(self, x):
        from ... import ops
        lin, drop = self.mlp[0][0], self.mlp[0][1]
        B = x.shape[0]
        y = ops.chan_mix(x.reshape(1, -1, x.shape[-1]).transpose(1, 2).contiguous(), lin.weight,
                         sink=(lin.weight, ...))   # [1,out,rows]
        y = y.transpose(1, 2).reshape(*x.shape[:-1], -1)
        if lin.bias is not None:
            y = y + lin.bias
        return expmap0(drop(y))
