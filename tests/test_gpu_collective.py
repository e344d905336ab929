"""The RCCL leg of the multi-GPU path on real hardware, every GPU test run (the reference spawns one rank per
GPU, base.py:61-65, and gathers results through lib/utils/comm.py:60-81): `bench.py --gpus 1 --force-collective` runs the
REAL rank body -- `init_process_group("nccl")` (= RCCL), the all-gather of the decoded per-video rows after every step,
barrier + max-over-ranks timing -- with world size 1.  More ranks are rehearsed on gloo (tests/test_dist_gloo.py); the
N > 1 hardware run is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_rank_body_with_rccl_all_gather_on_one_gpu(device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak"
    pr = out["per_rank"]
    assert pr["backend"] == "nccl" and pr["gathered_equals_local"] is True
    assert len(pr["ms_per_step"]) == 1 and len(pr["gather_ms"]) == 1
    assert 0.0 < pr["gather_ms"][0] < pr["ms_per_step"][0]          # the collective ran (HIP events) inside the step
    assert pr["units_per_step"] == [16 * 992]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
