"""f3, second half: the greedy association with its trajectory IoUs batched on the device
(`tspn_traj_iou_tail_f64`, one launch per segment) against the reference's own run (golden g9,
lib/modeling/association.py:117-175) and against the host IoU (`association._cubic_iou_1x1`, the float32 / float64
rounding recipe of lib/modeling/trajectory.py:85-141)."""
import copy

import numpy as np
import pytest
import torch

import cases
from test_association import check, unpack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L", [1, 7, 8, 9, 30, 127, 128, 129, 300, 1100])
def test_traj_iou_tail_bit_exact_vs_host_recipe(tspn, device, L):
    """Every (trajectory, tracklet) pair of one launch equals the one-by-one host evaluation to the bit: float32
    intersections in frame order, float64 areas in numpy's pairwise summation order (block of 8 interleaved partial
    sums up to 128 frames, split halves above), float64 quotient stored as float32.  Non-integer boxes on purpose."""
    A = tspn.association
    rs = np.random.RandomState(100 + L)
    U, N = 19, 13

    def boxes(n):
        xy = rs.uniform(0, 300, size=(n, L, 2))
        wh = rs.uniform(5, 250, size=(n, L, 2))
        return np.concatenate([xy, xy + wh], axis=2)

    a, b = boxes(U), boxes(N)
    a[3] = b[5] + rs.uniform(-2, 2, size=(L, 4))           # a near-identical pair (IoU close to 1)
    a[4, :, :2] += 5000                                     # a disjoint one (IoU exactly 0)
    ln = rs.randint(0, L + 1, size=U).astype(np.int32)
    ln[0], ln[1] = L, 0
    got = tspn.ops.traj_iou_tail(torch.from_numpy(a).to(device), torch.from_numpy(ln).to(device),
                                 torch.from_numpy(b).to(device)).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (U, N)
    ref = np.zeros((U, N), dtype=np.float32)
    for u in range(U):
        for n in range(N):
            if ln[u]:
                ref[u, n] = A._cubic_iou_1x1(a[u, :ln[u]], b[n, :ln[u]])
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert got[4].max() == 0 and got[1].max() == 0 and (ln[3] == 0 or got[3, 5] > 0.9)


def test_traj_iou_tail_argument_checks(tspn, device):
    a = torch.zeros((2, 5, 4), dtype=torch.float64, device=device)
    ln = torch.zeros((2,), dtype=torch.int32, device=device)
    with pytest.raises(TypeError):
        tspn.ops.traj_iou_tail(a.float(), ln, a)
    with pytest.raises(ValueError):
        tspn.ops.traj_iou_tail(a, ln, a[:, :4].contiguous())
    with pytest.raises(ValueError):
        tspn.ops.traj_iou_tail(a, ln[:1], a)
    with pytest.raises(RuntimeError):
        tspn.ops.traj_iou_tail(a.cpu(), ln.cpu(), a.cpu())
    assert tspn.ops.traj_iou_tail(a[:0], ln[:0], a).shape == (0, 2)


@pytest.mark.parametrize("cap", [100, 25])
def test_device_association_reproduces_the_reference_run_g9(tspn, device, cap):
    """Golden g9 = the reference's own greedy_relational_association: reproduced exactly (triplets, scores, durations,
    every box) with the IoUs coming from the batched launches."""
    g = cases.load("g9_association.npz")
    rels, trajs = cases.g9_scenario()
    stats = {}
    res = tspn.association.greedy_relational_association(None, copy.deepcopy(rels), max_traj_num_in_clip=cap,
                                                         trajectories=trajs, device=device, stats=stats)
    check(res, g, cap)
    assert stats["segments"] == 5 and 4 <= stats["iou_launches"] < stats["iou_lookups"]


def test_device_association_equals_host_at_a_larger_scale(tspn, device):
    """12 segments x 32 tracklets x 120 predictions (same generator as g9): device-batched IoUs == host IoUs."""
    rels, trajs = cases.g9_scenario(seed=23, n_seg=12, n_trk=32, n_pred=120)
    host = tspn.association.greedy_relational_association(None, copy.deepcopy(rels), trajectories=trajs)
    stats = {}
    dev = tspn.association.greedy_relational_association(None, copy.deepcopy(rels), trajectories=trajs, device=device,
                                                         stats=stats)
    a, b = unpack(host), unpack(dev)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert (a["sub_traj_len"] > 30).sum() > 20            # the scenario does merge across segments
    assert stats["iou_launches"] < stats["iou_lookups"]


def test_device_association_ragged_segment_raises_like_the_reference(tspn, device):
    """A trajectory that outlasts the current segment meets a shorter tracklet: trajectory.py:89 asserts; so does the
    host mirror; the device path raises when (and only when) that pair is looked at."""
    A = tspn.association
    box = [[0, 0, 50, 60]] * 60
    rels = [(("v", 0, 60), ([(np.array(0.9), np.array([1, 2, 3]), np.array([0, 1]))], None, None)),
            (("v", 15, 45), ([(np.array(0.8), np.array([1, 2, 3]), np.array([0, 1]))], None, None))]
    trajs = {("v", 0, 60): [box, box], ("v", 15, 45): [box[:30], box[:30]]}
    with pytest.raises(AssertionError):
        A.greedy_relational_association(None, copy.deepcopy(rels), trajectories=trajs)
    with pytest.raises(AssertionError):
        A.greedy_relational_association(None, copy.deepcopy(rels), trajectories=trajs, device=device)
    rels[1][1][0][0] = (np.array(0.8), np.array([9, 9, 9]), np.array([0, 1]))    # another triplet: never compared
    out = A.greedy_relational_association(None, copy.deepcopy(rels), trajectories=trajs, device=device)
    assert len(out) == 2


def test_association_worker_process_gives_the_same_relations(tspn, device):
    """Round 6: `AssociationWorker` runs `greedy_relational_association` for whole videos in a worker process (its own HIP
    context, a high-priority stream for the IoU tables; stale rows refreshed on the host) -- the relations must equal the
    in-process host result, from the tuple form and from the array form of the hand-over, and a failing job must
    come back as an error, not kill the worker."""
    import copy
    A = tspn.association
    rels, trajs = cases.g9_scenario(seed=31, n_seg=9, n_trk=16, n_pred=60)
    want = A.greedy_relational_association(None, copy.deepcopy(rels), max_traj_num_in_clip=60, trajectories=trajs)
    segs = [index for index, _ in rels]
    arr = {"scores": [np.array([p[0] for p in pr[0]]) for _, pr in rels],
           "triplets": [np.stack([p[1] for p in pr[0]]) for _, pr in rels],
           "pairs": [np.stack([p[2] for p in pr[0]]) for _, pr in rels], "boxes": [trajs[i] for i in segs]}
    with A.AssociationWorker(device=str(device), max_traj_num_in_clip=60) as w:
        w.submit("tuples", copy.deepcopy(rels), trajs)
        w.submit_arrays("arrays", segs, arr["scores"], arr["triplets"], arr["pairs"], arr["boxes"])
        w.submit("broken", [(("v", 0, 30), ([], None, None))], {})
        key, out, ms, stats = w.result()
        assert key == "tuples" and out == want and stats["iou_launches"] >= 1 and stats["iou_host_rows"] >= 1
        key, out, ms, stats = w.result()
        assert key == "arrays" and out == want
        with pytest.raises(RuntimeError, match="KeyError"):
            w.result()
        w.submit("again", copy.deepcopy(rels), trajs, return_relations=False)
        assert w.result()[1] == len(want)
