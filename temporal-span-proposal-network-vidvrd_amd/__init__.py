"""MI355X-native TSPN relation-scoring hot path (gfx950).

Drop-in for the reference's `BaseModel` (lib/modeling/model.py) on the path
named by BASELINE.json: N^2 pair builder -> temporal context encoder ->
relationness / span-regression / predicate heads, as hand-written HIP kernels
behind a C ABI (include/tspn_mi355x.h), with a Python host mirroring the
reference's module interface.

The directory name is not a valid Python identifier; import it through the
repo-root shim:  `import tspn_mi355x`  (see tspn_mi355x.py), or
`importlib.import_module("temporal-span-proposal-network-vidvrd_amd")`.
"""
from . import _abi, association, config, dataset, hashrng, ops, predict, synth  # noqa: F401
from .config import Cfg, default_cfg, load_cfg, merge_from_file  # noqa: F401
from .model import (BaseModel, DPN, DPNHead, PPN, PPNHead, RelOIPool, RelPN,  # noqa: F401
                    RelationPredictor, TemporalProposals, make_relpn)
from .pair_list import PairList, TargetList  # noqa: F401
from .sampler import BalancedPositiveNegativePairSampler  # noqa: F401
from .anchor_generator import AnchorGenerator, generate_anchors, make_anchor_generator  # noqa: F401
from . import dist  # noqa: F401
from . import roi_head  # noqa: F401
from .roi_head import Res5RoIHead, ResNetC4  # noqa: F401

__version__ = "0.1.0"
