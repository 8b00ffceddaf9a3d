"""Mirror of hpcs.nn (hpcs/nn/__init__.py:4-9)."""
from torch.nn import Dropout, Linear, Sequential


def MLP(channels, bias=True, negative_slope=0.0, dropout=0.0):
    """Stack of Linear(+Dropout) blocks -- no non-linearity, exactly like the reference's MLP."""
    blocks = [Sequential(Linear(channels[i - 1], channels[i], bias=bias), Dropout(dropout))
              for i in range(1, len(channels))]
    return Sequential(*blocks)
