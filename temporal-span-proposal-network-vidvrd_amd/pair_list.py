"""Input container types of `BaseModel.forward`.

Interface-compatible with the reference's `PairList` / `TargetList`
(lib/dataset/list_pair.py:3-57, lib/dataset/list_target.py:3-57): a primary
tensor plus a dict of named extra fields, `.to(device)` moving the tensor and
every field that itself has a `.to` (numpy fields stay on the host, exactly as
in the reference — SURVEY.md §8a row a14).  Reference objects are accepted by
the model as they are (duck typing); these classes exist so the path has no
import dependency on the reference.

Fields used by the hot path:
  baseline path : features [P,F], 'track_cls_logits' [N,35], 'tracklet_pairs' [P,2],
                  'num_tracklets'
  temporal path : 'tracklet_feats' [N,T,D], 'tracklet_boxes' [N,T,4] (+ the above);
                  `features` may instead hold a materialised [P,C,T] pair tensor.
"""
import torch


class _FieldList:
    _primary = "data"

    def __init__(self, primary):
        object.__setattr__(self, self._primary, primary)
        self.extra_fields = {}

    # -- field access -----------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def _copy_extra_fields(self, other):
        self.extra_fields.update(other.extra_fields)

    # -- tensor-like ------------------------------------------------------
    def _primary_value(self):
        return getattr(self, self._primary)

    def _derive(self, primary, field_fn):
        out = type(self)(primary)
        for name, value in self.extra_fields.items():
            out.add_field(name, field_fn(value))
        return out

    def to(self, device, non_blocking=False):
        """Reference semantics (list_pair.py:28-31): the primary tensor and every field with a `.to` move, numpy
        fields stay.  `non_blocking=True` (build extension) makes the copies of PINNED host tensors asynchronous on the
        current stream, which is what a prefetching loader wants."""
        if not non_blocking:
            return self._derive(self._primary_value().to(device),
                                lambda v: v.to(device) if hasattr(v, "to") else v)
        mv = lambda v: v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else (v.to(device) if hasattr(v, "to") else v)  # noqa: E731
        return self._derive(mv(self._primary_value()), mv)

    def pin_memory(self):
        """Page-locked copies of the host tensors (primary + tensor fields): `torch.utils.data.DataLoader(pin_memory=
        True)` calls this on the batches its workers produce (it looks for a `pin_memory` method on custom types), in
        its own thread.  `BaseModel.forward` DMAs pinned tracklet features straight from where they are."""
        pin = lambda v: v.pin_memory() if isinstance(v, torch.Tensor) and not v.is_cuda else v  # noqa: E731
        return self._derive(pin(self._primary_value()), pin)

    def __getitem__(self, item):
        return self._derive(self._primary_value()[item], lambda v: v[item])

    def __len__(self):
        return self._primary_value().shape[0]

    def copy_with_fields(self, fields, skip_missing=False):
        out = type(self)(self._primary_value())
        if not isinstance(fields, (list, tuple)):
            fields = [fields]
        for name in fields:
            if self.has_field(name):
                out.add_field(name, self.get_field(name))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(name, self))
        return out


class PairList(_FieldList):
    """Per-segment pair features + fields (reference lib/dataset/list_pair.py)."""
    _primary = "features"

    def __repr__(self):
        return "PairList(num_feats={})".format(len(self))

    @classmethod
    def from_tracklets(cls, tracklet_feats, tracklet_boxes=None, track_cls_logits=None,
                       tracklet_pairs=None):
        """Temporal-path sample: tracklet tensors instead of precomputed pair features.

        `features` is an empty [P,0] placeholder; P = N(N-1) ordered pairs
        (i-major, predict.py:133-140) unless `tracklet_pairs` is given.
        """
        n = tracklet_feats.shape[0]
        p = n * (n - 1) if tracklet_pairs is None else len(tracklet_pairs)
        out = cls(torch.empty((p, 0), dtype=tracklet_feats.dtype, device=tracklet_feats.device))
        out.add_field("tracklet_feats", tracklet_feats)
        if tracklet_boxes is not None:
            out.add_field("tracklet_boxes", tracklet_boxes)
        if track_cls_logits is not None:
            out.add_field("track_cls_logits", track_cls_logits)
        if tracklet_pairs is not None:
            out.add_field("tracklet_pairs", tracklet_pairs)
        out.add_field("num_tracklets", n)
        return out


class TargetList(_FieldList):
    """Per-segment multi-hot predicate targets [P,K] (reference lib/dataset/list_target.py)."""
    _primary = "target"

    def __repr__(self):
        return "TargetList(num_targets={})".format(len(self))
