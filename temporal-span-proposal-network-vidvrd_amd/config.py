"""Configuration surface of the hot path.

The reference reads a yacs `CfgNode` by attribute access only
(lib/modeling/model.py:10-18, relpn/ppn.py:13-23, relpn/dpn.py:76-81,
relpn/rel_nms.py:11), so any attribute-style mapping is a drop-in.  yacs is not
a dependency here: `Cfg` is a minimal attribute dict, `default_cfg()` restates
the values of lib/config/defaults.py:3-74 that the hot path reads, and
`load_cfg(path)` overlays a plain YAML file such as configs/baseline.yaml
(always with `yaml.safe_load`: the reference also ships a pickled-object YAML,
configs/baseline_config.yaml, that must never be loaded unsafely).
A real yacs CfgNode can be passed to `BaseModel` directly as well.
"""
import copy

import yaml


class Cfg(dict):
    """dict with attribute access, nested."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self):
        return copy.deepcopy(self)


def _wrap(obj):
    if isinstance(obj, dict):
        return Cfg({k: _wrap(v) for k, v in obj.items()})
    return obj


_DEFAULTS = {
    "MODEL": {"NAME": "baseline"},
    "DATASET": {"TRAIN_BATCH_SIZE": 1024, "TEST_BATCH_SIZE": 1, "TRAIN_NUM_WORKERS": 0,
                "TEST_NUM_WORKERS": 4, "LOGIT_ONLY": False, "USE_GT_OBJ_TRAJS": False},
    "PREDICT": {"OBJECT_NUM": 35, "PREDICATE_NUM": 132, "TOPK_PER_PAIR": 20, "TOPK_PER_SEG": 200,
                "FEATURE_DIM": 11070},
    "RELPN": {
        "OBJECT_DIM": 1024,
        "USE_PPN": True,
        "USE_DPN": True,
        # build extension: PPN and the top-k triplet decode run on a second HIP stream under the encoder of the same
        # forward (they depend on the class / predicate logits only) and join the caller's stream before returning
        "OVERLAP_TAIL": True,
        "PPN": {"NUM_PAIR_PROPOSALS": 256, "IN_CHANNELS": 35, "HIDDEN_CHANNELS": 64,
                "OUT_CHANNELS": 35, "BATCH_SIZE_PER_SEGMENT": 256, "POSITIVE_FRACTION": 0.5},
        "DPN": {"NUM_DURATION_PROPOSALS": 64, "DPN_ONLY": False, "IN_CHANNELS": 1024,
                "NUM_ANCHORS_PER_LOCATION": 4, "ANCHOR_SIZES": 35, "ANCHOR_STRIDE": 132,
                # build extension (not in the reference's defaults.py): RelOIPool restricted to each
                # pair's top temporal span instead of the whole segment (model.py:68-73 is a stub)
                "POOL_TOP_SPAN": False,
                # build extension: forward also returns the relative box geometry [P,8,T] of every pair
                # (TemporalProposals.geom, the bbox half of the pair builder); nothing downstream reads it
                "PAIR_GEOMETRY": False,
                # build extension: algorithm of the k=3 temporal conv on the GPU (both exact fp32 MFMA):
                # "auto" = Winograd F(6,3) where the shape allows it (D % 32 == 0; 4/9 of the direct MFMA work,
                # error bound in DESIGN.md §4), else the direct taps; "direct" = always the direct taps
                "CONV_ALGO": "auto",
                # build extension (round 6): accuracy guard of "auto".  Every fused pass that runs F(6,3) recomputes
                # CONV_CHECK_ROWS output rows (x up to 24 columns, the sextet with the largest |x| among them) in float64
                # on the GPU; when the measured error exceeds CONV_TOL the model warns once and uses the direct taps from
                # its next call on (model.BaseModel._winograd).  0 rows = guard off.
                "CONV_TOL": 1e-4,
                "CONV_CHECK_ROWS": 128,
                # build extension: tracklet features handed over in HOST memory (predict.py:50-57) go to the device in
                # chunks of this many videos, pipelined under the encoder (model._HostPipeline)
                "HOST_CHUNK_VIDEOS": 4},
    },
    "ETC": {"RANDOM_SEED": 0, "MODEL_DUMP_FILE": "baseline_weights_epoch_100.pt"},
}


def default_cfg():
    """Hot-path subset of the reference defaults (lib/config/defaults.py:3-74)."""
    return _wrap(copy.deepcopy(_DEFAULTS))


def _merge(dst, src, path=""):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v, f"{path}{k}.")
        else:
            dst[k] = _wrap(v)


def merge_from_file(cfg, path):
    with open(path, "r") as fh:
        data = yaml.safe_load(fh)
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a YAML mapping")
    _merge(cfg, data)
    return cfg


def load_cfg(path=None, **overrides):
    """default_cfg() overlaid with a YAML file and dotted-key overrides
    (e.g. load_cfg('configs/baseline.yaml', **{'RELPN.USE_DPN': True}))."""
    cfg = default_cfg()
    if path is not None:
        merge_from_file(cfg, path)
    for dotted, value in overrides.items():
        node = cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node.setdefault(k, Cfg())
        node[keys[-1]] = value
    return cfg
