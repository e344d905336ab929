#!/usr/bin/env python3
"""Greedy relational association at VidOR scale: 59 overlapping 30-frame segments x 64 tracklets x 200 predictions
(the decoded top-200 of every segment, predict.py:106-116), three ways:

  reference-order  oracle.greedy_association: the reference's statement order (association.py:117-175) - re-sorts the
                   previous segment's relations before every prediction, one numpy IoU per (prediction, candidate)
  host             association.greedy_relational_association: sort and same-triplet grouping once per segment, host IoU
  device           the same with device=: every (live trajectory, tracklet) IoU of a segment in ONE launch of
                   tspn_traj_iou_tail_f64, stale rows recomputed in one launch

All three must give identical relations.  Prints one JSON line; `--skip-reference` drops the slow first variant.
    python tools/bench_association.py [--segments 59 --tracklets 64 --predictions 200 --runs 3]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=59)
    ap.add_argument("--tracklets", type=int, default=64)
    ap.add_argument("--predictions", type=int, default=200)
    ap.add_argument("--cap", type=int, default=200, help="max_traj_num_in_clip (reference default 100)")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--no-device", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import cases
    import oracle
    import tspn_mi355x as tspn

    rels, trajs = cases.g9_scenario(seed=31, n_seg=args.segments, n_trk=args.tracklets, n_pred=args.predictions)

    def timed(fn, runs):
        best, out = None, None
        for _ in range(runs):
            r = copy.deepcopy(rels)
            t0 = time.perf_counter()
            out = fn(r)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, out

    def key(res):
        return [(tuple(r["triplet"]), r["score"], tuple(r["duration"]), len(r["sub_traj"]), len(r["obj_traj"]),
                 tuple(np.asarray(r["sub_traj"]).ravel().tolist()[:8])) for r in res]

    line = {"segments": args.segments, "tracklets": args.tracklets, "predictions_per_segment": args.predictions,
            "max_traj_num_in_clip": args.cap, "cores": len(os.sched_getaffinity(0))}
    A = tspn.association
    t_host, res_host = timed(lambda r: A.greedy_relational_association(None, r, max_traj_num_in_clip=args.cap,
                                                                       trajectories=trajs), args.runs)
    line["host_ms"] = t_host * 1e3
    line["relations"] = len(res_host)
    line["extended"] = int(sum(len(r["sub_traj"]) > 30 for r in res_host))
    if not args.skip_reference:
        t_ref, res_ref = timed(lambda r: oracle.greedy_association(r, trajs, max_traj_num_in_clip=args.cap), 1)
        line["reference_order_ms"] = t_ref * 1e3
        assert key(res_ref) == key(res_host), "host association differs from the reference-order restatement"
    if not args.no_device and torch.cuda.is_available():
        stats = {}
        A.greedy_relational_association(None, copy.deepcopy(rels), max_traj_num_in_clip=args.cap, trajectories=trajs,
                                        device="cuda")     # warm-up: library load, allocator
        t_dev, res_dev = timed(lambda r: A.greedy_relational_association(None, r, max_traj_num_in_clip=args.cap,
                                                                         trajectories=trajs, device="cuda", stats=stats),
                               args.runs)
        assert key(res_dev) == key(res_host), "device association differs from the host one"
        line["device_ms"] = t_dev * 1e3
        line.update({k: v for k, v in stats.items()})
    print(json.dumps(line))


if __name__ == "__main__":
    main()
