"""Mirror of the hot-path part of hpcs/utils/data.py: `to_categorical` (:24-29), the one-hot category code
`ShapeNetHypHC._forward` feeds the extractor (hpcs/models/shapenet_hyp_hc.py:84).  The reference builds the one-hot on the
host (`torch.eye(C)[y.cpu().numpy()]`, then `.cuda()`): a device->host->device hop per step; here it is formed on y's
device.  The dataset paths and `pc_normalize` of that file are host I/O: not mirrored,
`hpcs_amd.install_as` falls back to the reference's file for them."""
import torch


_EYE = {}


def to_categorical(y, num_classes):
    """1-hot encodes an integer tensor [...]-> [..., num_classes] float32, on the device of y: rows of a cached identity
    (`torch.eye(C)[y]`, the reference's own expression) -- one gather launch where F.one_hot + .float() took four."""
    key = (int(num_classes), y.device)
    eye = _EYE.get(key)
    if eye is None:
        eye = _EYE[key] = torch.eye(int(num_classes), dtype=torch.float32, device=y.device)
    return eye[y.long()]


def remap_labels(y_true):
    """Relabels to 0..K-1 in sorted order of the labels present (data.py:31-35)."""
    return torch.unique(y_true, return_inverse=True)[1]
