"""1-D temporal anchors — host mirror of the reference's `AnchorGenerator`
(lib/modeling/relpn/anchor_generator.py:31-104; only its unimportable `dpn_anchor.py` uses it).

An anchor is a (start, end) window: width `sizes[a]` centred on every `anchor_stride`-th frame of the
time axis, `0 .. T` inclusive, location-major / size-minor — the candidate order `c = t*A + a` that
the GPU span decode (`tspn_decode_spans_f32`, stride-1 grid) uses.  Works on numpy >= 1.24 (the
reference needs the removed `np.float`)."""
import numpy as np
import torch
from torch import nn

__all__ = ["AnchorGenerator", "generate_anchors", "make_anchor_generator"]


def generate_anchors(stride=8, sizes=(4, 8, 16)):
    """Cell anchors around frame 0: rows (-size/2, +size/2), float64 tensor [A,2]
    (anchor_generator.py:67-104: `stride * (size / stride)` wide)."""
    widths = float(stride) * (np.asarray(sizes, dtype=np.float64) / float(stride))
    return torch.from_numpy(np.stack([0.0 - 0.5 * widths, 0.0 + 0.5 * widths], axis=1))


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(4, 8, 16), anchor_stride=8):
        super().__init__()
        self.stride = anchor_stride
        self.register_buffer("cell_anchors_0", generate_anchors(anchor_stride, sizes).float())

    @property
    def cell_anchors(self):
        return [self.cell_anchors_0]

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def grid_anchors(self, time_width):
        """list with one float32 tensor [S*A, 2], S = len(arange(0, T+1, stride))
        (anchor_generator.py:48-59)."""
        out = []
        for base in self.cell_anchors:
            shifts = torch.arange(0, time_width + 1, step=self.stride, dtype=torch.float32, device=base.device)
            out.append((shifts.view(-1, 1, 1) + base.reshape(1, -1, 1)).reshape(-1, 2))
        return out

    def forward(self, rel_feats):
        return self.grid_anchors(rel_feats.shape[2])   # N x C x T (time last)


def make_anchor_generator(cfg):
    """cfg.RELPN.DPN.ANCHOR_SIZES / ANCHOR_STRIDE.  The reference's factory asserts `len()` of an int
    default and cannot run (anchor_generator.py:107-123, defaults.py:66-67); here a scalar stride is
    accepted and a scalar ANCHOR_SIZES falls back to the example of its own comment block,
    sizes (15, 30, 45, 60)."""
    sizes = cfg.RELPN.DPN.ANCHOR_SIZES
    stride = cfg.RELPN.DPN.ANCHOR_STRIDE
    if isinstance(stride, (list, tuple)):
        if len(stride) != 1:
            raise ValueError("should have a single ANCHOR_STRIDE")
        stride = stride[0]
    if not isinstance(sizes, (list, tuple)):
        sizes = (15, 30, 45, 60)
    return AnchorGenerator(tuple(sizes), stride)
