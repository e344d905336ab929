"""Greedy relational association (host post-processing): the product's host module and
the oracle's restatement against the golden produced by the reference's own
`greedy_relational_association` (tests/golden/make_golden.py:g9_association)."""
import copy

import numpy as np
import pytest

import cases
import oracle


def unpack(res, key_traj=("sub_traj", "obj_traj")):
    out = {"triplet": np.array([r["triplet"] for r in res], dtype=np.int64),
           "score": np.array([r["score"] for r in res], dtype=np.float64),
           "duration": np.array([r["duration"] for r in res], dtype=np.int64)}
    for side in key_traj:
        out[side + "_len"] = np.array([len(r[side]) for r in res], dtype=np.int64)
        out[side] = np.array([b for r in res for b in np.asarray(r[side]).reshape(-1, 4)], dtype=np.float64).reshape(-1, 4)
    return out


def check(res, g, cap):
    got = unpack(res)
    for k, v in got.items():
        np.testing.assert_array_equal(v, g[f"cap{cap}_{k}"], err_msg=k)


@pytest.mark.parametrize("cap", [100, 25])
def test_host_association_matches_reference(tspn, cap):
    g = cases.load("g9_association.npz")
    rels, trajs = cases.g9_scenario()
    res = tspn.association.greedy_relational_association(None, copy.deepcopy(rels), max_traj_num_in_clip=cap,
                                                         trajectories=trajs)
    check(res, g, cap)
    # the same through a callable provider handing over traj_cls-style dicts
    def provider(vid, fs, fe):
        return [{"pstart": fs, "pend": fe, "rois": b.tolist(), "score": 0.5, "category": 1, "classeme": [0.0]}
                for b in trajs[(vid, fs, fe)]]
    res2 = tspn.association.greedy_relational_association(None, copy.deepcopy(rels), max_traj_num_in_clip=cap,
                                                          trajectories=provider)
    check(res2, g, cap)


@pytest.mark.parametrize("cap", [100, 25])
def test_oracle_association_matches_reference(cap):
    g = cases.load("g9_association.npz")
    rels, trajs = cases.g9_scenario()
    check(oracle.greedy_association(copy.deepcopy(rels), trajs, max_traj_num_in_clip=cap), g, cap)


def test_association_quirks_and_names(tspn):
    """Known answers on a two-segment toy: merge averages the overlap and appends the rest; a relation
    opened after the first segment starts its confidence list with 1 (reference association.py:166);
    relations sharing a tracklet see each other's merge (in-place trajectories); names come from the dataset."""
    A = tspn.association
    box = lambda x: [[x, 0, x + 50, 60]] * 30          # noqa: E731
    trajs = {("v", 0, 30): [box(0), box(200)], ("v", 15, 45): [box(2), box(204), box(600)]}
    tri = np.array([1, 2, 3])
    rels = [(("v", 15, 45), ([(np.array(0.9), tri, np.array([0, 1])), (np.array(0.4), np.array([4, 4, 4]), np.array([2, 1]))], None, None)),
            (("v", 0, 30), ([(np.array(0.5), tri, np.array([0, 1])), (np.array(0.3), np.array([1, 0, 3]), np.array([0, 1]))], None, None))]

    class Names:
        def get_object_name(self, i):
            return f"obj{int(i)}"

        def get_predicate_name(self, i):
            return f"pred{int(i)}"

    res = A.greedy_relational_association(Names(), rels, trajectories=trajs)
    assert [r["triplet"] for r in res] == [["obj1", "pred2", "obj3"], ["obj1", "pred0", "obj3"], ["obj4", "pred4", "obj4"]]
    assert res[0]["duration"] == [0, 45] and res[0]["score"] == pytest.approx(0.7)
    assert len(res[0]["sub_traj"]) == 45
    assert res[0]["sub_traj"][0] == (0.0, 0.0, 50.0, 60.0) and res[0]["sub_traj"][20] == (1.0, 0.0, 51.0, 60.0)
    assert res[0]["sub_traj"][40] == (2.0, 0.0, 52.0, 60.0)
    # the second first-segment relation shares both tracklets: its trajectories grew too, its duration did not
    assert len(res[1]["sub_traj"]) == 45 and res[1]["duration"] == [0, 30] and res[1]["score"] == pytest.approx(0.3)
    # opened in the second segment: confidence 1, not 0.4
    assert res[2]["score"] == 1.0 and res[2]["duration"] == [15, 45]
    assert A._traj_iou(A.Track(0, 10, box(0)[:10]), A.Track(10, 20, box(0)[:10])) == 0   # no common frame
    with pytest.raises(ValueError):
        A.Track(0, 5, box(0))


def test_association_empty_and_capped_inputs(tspn):
    A = tspn.association
    assert A.greedy_relational_association(None, [], trajectories={}) == []
    box = [[0, 0, 10, 10]] * 30
    rels = [(("v", 0, 30), ([], None, None)), (("v", 15, 45), ([(np.array(0.5), np.array([1, 1, 1]), np.array([0, 1]))], None, None))]
    out = A.greedy_relational_association(None, rels, trajectories={("v", 0, 30): [box, box], ("v", 15, 45): [box, box]})
    assert len(out) == 1 and out[0]["duration"] == [15, 45] and out[0]["score"] == 1.0   # opened after segment 0
    many = [(np.array(0.01 * i), np.array([i % 3, 0, 1]), np.array([0, 1])) for i in range(10)]
    out = A.greedy_relational_association(None, [(("v", 0, 30), (many, None, None))], max_traj_num_in_clip=4,
                                          trajectories={("v", 0, 30): [box, box]})
    assert [round(r["score"], 2) for r in out] == [0.09, 0.08, 0.07, 0.06]
    assert A.load_trajectories("no-such-video", 0, 30, root="/nonexistent") == []


def test_numpy_pairwise_summation_order_restated():
    """The device kernel `traj_iou_tail_f64_kernel` sums the float64 box areas in numpy's pairwise order so that it
    equals `np.sum` (trajectory.py:110-123) to the bit.  This pins that order on the CPU: the restatement below (the
    algorithm csrc/tspn_iou.hip implements) equals np.sum exactly for every length 0..1300 on ill-conditioned data."""
    def block(a):
        n = len(a)
        if n < 8:
            r = 0.0
            for v in a:
                r += v
            return r
        r = [float(v) for v in a[:8]]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] += a[i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for v in a[i:]:
            res += v
        return res

    def pairwise(a):
        n = len(a)
        if n <= 128:
            return block(a)
        n2 = n // 2
        n2 -= n2 % 8
        return pairwise(a[:n2]) + pairwise(a[n2:])

    rs = np.random.RandomState(5)
    for n in list(range(0, 140)) + [255, 256, 257, 300, 511, 777, 1024, 1100, 1300]:
        a = (rs.uniform(1, 2, size=n) * 10.0 ** rs.randint(-3, 6, size=n)).astype(np.float64)
        assert float(np.sum(a)) == pairwise([float(v) for v in a]), n


def test_vectorised_host_iou_rows_equal_the_pairwise_form_bit_for_bit():
    """`_cubic_iou_1xn` (round 6: the rows of a segment's IoU table that merges have made stale are refreshed on the host,
    one trajectory against all tracklets at once) must reproduce `_cubic_iou_1x1` -- the roundings of
    lib/modeling/trajectory.py:85-141 -- to the bit, on integer boxes and on the half-integer boxes merges produce."""
    import tspn_mi355x as tspn
    A = tspn.association
    rs = np.random.RandomState(5)
    for k in (1, 7, 15, 30):
        for scale in (1.0, 0.5, 0.25):
            xy = np.round(rs.uniform(0, 600, size=(33, k, 2)) / scale) * scale
            wh = np.round(rs.uniform(5, 300, size=(33, k, 2)) / scale) * scale
            b = np.concatenate([xy, xy + wh], axis=2)
            a = b[0] + np.round(rs.uniform(-20, 20, size=(k, 4)) / scale) * scale
            a[:, 2:] = np.maximum(a[:, 2:], a[:, :2] + 1)
            far = b[1] + 5000.0                                   # no overlap at all -> exactly 0
            row = A._cubic_iou_1xn(a, np.concatenate([b, far[None]]))
            want = np.array([A._cubic_iou_1x1(a, bb) for bb in np.concatenate([b, far[None]])], dtype=np.float32)
            assert row.dtype == np.float32 and row.shape == (34,)
            np.testing.assert_array_equal(row.view(np.uint32), want.view(np.uint32))
            assert row[-1] == 0.0 and row[0] > 0.0
