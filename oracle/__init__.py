"""CPU oracle for the TSPN relation-scoring hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It restates, in plain torch/numpy on the CPU, the algorithm of the reference's
hot path (`/root/reference/lib/modeling/{model.py,relpn/*.py,trajectory.py}`,
`lib/dataset/vrdataset.py:_feature_preprocess`, `lib/modeling/predict.py`
decode).  Every function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it — as the checker, never as the thing measured or
shipped.  The product package (`temporal-span-proposal-network-vidvrd_amd/`)
never imports it and has no CPU fallback: it raises when the HIP library or a
GPU is missing.

Parity pinning: the reference has no tests / golden vectors
(SURVEY.md §4).  The oracle is pinned against outputs of the reference's own
Python modules, imported from `/root/reference` in the build container by
`tests/golden/make_golden.py`; those outputs are committed as
`tests/golden/*.npz`.  Functions with no live reference counterpart (pair
gather, relationness head, RelOIPool over time, span decode) are marked
"build-defined" below: their parity is pinned only by composition of the
reference-pinned pieces.
"""
from .tspn_oracle import *  # noqa: F401,F403
from . import roi_head_oracle  # noqa: F401  (RoI feature head; parity unpinned, see its header)
