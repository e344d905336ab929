#!/usr/bin/env python3
"""Headline benchmark: tracklet-pairs/sec scored at BASELINE.json cfg2
(synthetic VidVRD shape: N=32 tracklets, T=150 frames, D=2048 RoI dims, fp32).

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 needs nothing else: when no launcher has set WORLD_SIZE, this script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a CHILD process
before anything here has touched the GPU (the reference spawns its ranks itself too, base.py:61-65) and
exits with the child's code; under an external torch.distributed.run it just runs as the rank it is given.

Workloads (`--workload`):
  cfg2 (default, the headline) 16 videos per GPU per step of BASELINE.json configs[1]
  cfg4   the per-GPU shard of configs[3]: 512 cfg2-shaped videos over 8 GPUs = 64 videos per GPU per step
  cfg3   configs[2]: N=64, T=900, D=1024, bf16 operands, 4 videos per step
  cfg5   configs[4]: frames -> triplets at VidOR scale, one video (900 frames of 720p, 64 tracklets) per GPU per
         step: ResNet-101-C4 backbone (bf16 MFMA convs) -> res4 maps -> RoIAlign + res5 + mean = tracklet_feats
         [64,900,2048] -> BaseModel.forward (bf16 scorer) -> BaseModel.decode

The step runs THROUGH THE DROP-IN SURFACE by default: B device-resident `PairList.from_tracklets` per step ->
`BaseModel.forward(pair_list)` (reference lib/modeling/model.py:53-65, called as predict.py:57 calls it) ->
`BaseModel.decode` (predict.py:59-117).  `--ops-level` times the same kernels through `ops.forward_fused` with
caller-held workspace and outputs instead (the two agree within 1 %, profiles/r3/bench_via_model_vs_ops.md).

A "step" = one pass of the hot path over one batch of synthetic videos per GPU (inputs resident in HBM, a
rotation of `--batches` different batches): tracklet tensors -> [pair builder + temporal encoder +
relationness/span heads + RelOIPool + predicate head] (tspn_forward_fused_f32) + pair geometry [P,8,T] from the
boxes (tspn_pair_gather_f32) + PPN pair-matrix/top-k + top-k triplet decode (tspn_decode_topk_f32, the
reference's predict.py:66-106).  Videos shard across ranks (weak scaling, no collective in the forward); with
N>1 each step ends with ONE RCCL all-gather of the DECODED per-video results — top-200 (score, triplet, pair) +
top-256 pair proposals, 10.8 KB per video (`--gather logits` gathers the 524 KB of predicate logits per video
instead, as round 1 did).

Printed JSON (rank 0): see the task contract; extras:
  roofline     dominant kernel = the temporal conv of the tracklet projections (fp32 MFMA; Winograd
               F(6,3) by default, --conv direct for the direct taps); achieved = executed FLOP per launch /
               HIP-event time of that launch inside the timed steps (events recorded on the launch stream by
               the C ABI's hook); clock_mhz = shader clock sampled from the driver while the timed steps ran
               (box-to-box variance of `frac` is mostly the clock the chip holds under this kernel).
               cfg5: the backbone's convolutions as a whole (executed conv FLOP of one video / HIP-event time of
               the backbone inside the timed steps) against the dense bf16 MFMA peak.
  cpu_baseline the oracle on a FIXED bounded sample of the same workload on the cores this process may use
               (cfg2/cfg4: dense fp32 forward of 248 pairs, median of 5; cfg3: bf16 restatement of 96 pairs, one
               run; cfg5: backbone of 8 frames + RoI head of their 64 boxes each + scorer of 8 pairs, one run each,
               scaled to a video).

`--stub-gpu` (CPU rehearsal, used by tests/test_dist_gloo.py): the rank body of this file — rendezvous, warm-up,
barrier-bracketed timed loop, the decoded-result all-gather of every step, max-over-ranks time, the
`gathered[lo:hi] == local` check (lo, hi = this rank's block, ragged under `--total-videos`), the JSON line — with the GPU step replaced by recorded decoded
rows and `gloo` in place of RCCL.  Its JSON says `"stub": true`; it measures nothing.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
N_TRK, T_FRAMES, D_ROI, A_ANCH, K_PRED = 32, 150, 2048, 4, 132
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (no sparsity), v_mfma_f32_32x32x16_bf16
PEAK_CLOCK_MHZ = 2400.0  # the clock the peak figures are quoted at
# --workload cfg3: BASELINE.json configs[2] (VidOR long-clip shape, bf16 operands); not the headline
CFG3 = (64, 900, 1024)
# --workload cfg5: BASELINE.json configs[4] (frames -> triplets at VidOR scale): tracklets, frames, frame size
CFG5 = (64, 900, 720, 1280)
DPN_PRE = "relpn.duration_proposal_network.dpn_head."
PPN_PRE = "relpn.pair_proposal_network.ppn_head."
CPU_SUBJECTS = 8   # cpu_baseline sample (cfg2): all 31 objects of this many subjects
TOPK_PAIR, TOPK_SEG, TOPK_PPN = 20, 200, 256   # PREDICT.TOPK_PER_PAIR / TOPK_PER_SEG, PPN.NUM_PAIR_PROPOSALS
DEFAULT_STEPS = {"cfg2": (50, 5), "cfg4": (12, 2), "cfg3": (50, 5), "cfg5": (3, 1)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 50; cfg4 12; cfg5 3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 5; cfg4 2; cfg5 1)")
    ap.add_argument("--videos", type=int, default=None,
                    help="videos per GPU per step (default: 16 for cfg2, 64 for cfg4, 4 for cfg3, 1 for cfg5)")
    ap.add_argument("--total-videos", type=int, default=None,
                    help="videos per step over ALL ranks, block-sharded with dist.shard_range (ragged when it does not "
                         "divide: 509 over 8 ranks = 64,64,64,64,64,63,63,63); overrides --videos")
    ap.add_argument("--batches", type=int, default=2,
                    help="different input batches resident in HBM, rotated step by step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default cfg2 run on one GPU only: skip the short cfg2-direct / cfg4-shard / cfg3 / cfg5 legs that run AFTER the "
                         "timed region and are printed under \"secondary\"")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed runs of the cfg2 CPU baseline (median reported)")
    ap.add_argument("--conv", choices=["winograd6", "direct"], default="winograd6",
                    help="temporal-conv algorithm of the tracklet projections (both fp32 MFMA): Winograd F(6,3) "
                         "(default, what RELPN.DPN.CONV_ALGO = auto selects at this shape) or the direct taps")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 = headline (N=32,T=150,D=2048, fp32, 16 videos/GPU/step); cfg4 = its 64-videos-per-GPU "
                         "shard of the 512-video batch; cfg3 = N=64,T=900,D=1024 bf16 operands; cfg5 = frames -> "
                         "triplets (ResNet-101-C4 + RoI head + scorer) at VidOR scale, one video per step")
    ap.add_argument("--gather", choices=["decoded", "logits"], default="decoded",
                    help="payload of the N>1 result gather: decoded top-k results (default) or raw logits")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the RCCL result gather even with one rank (exercises the N>1 code path)")
    ap.add_argument("--overlap", action="store_true",
                    help="--ops-level only: pair geometry, PPN, decode and the result gather on a second HIP stream "
                         "behind the logits-ready event, under the encoder of the same step (measured on one GPU: "
                         "32.70 vs 32.65 ms per step, so the default keeps one stream)")
    ap.add_argument("--ops-level", action="store_true",
                    help="time ops.forward_fused with pre-allocated workspace / outputs instead of BaseModel.forward + "
                         "BaseModel.decode on PairLists (the default)")
    ap.add_argument("--host-inputs", choices=["off", "pageable", "pinned", "prefetch", "prefetch-pinned", "naive"], default="off",
                    help="cfg2/cfg3/cfg4 through BaseModel: the PairLists hold HOST tensors, as the reference's predict.py:50-57 "
                         "hands them (pageable = a plain DataLoader; pinned = DataLoader(pin_memory=True) / PairList."
                         "pin_memory()); forward pipelines upload, encoder and download over chunks of videos and returns "
                         "host results; prefetch[-pinned] = the same host batches through dataset.DevicePrefetcher (batch i+1 uploads "
                         "under batch i, forward sees device tensors); naive = round 3's path, one blocking .to(device) per "
                         "tensor.  The line then reports the PCIe-INCLUSIVE rate (never the headline value)")
    ap.add_argument("--host-chunk", type=int, default=None, help="--host-inputs: videos per pipelined chunk (default 4)")
    ap.add_argument("--frames", type=int, default=None, help="cfg5: frames per video (default 900)")
    ap.add_argument("--tracklets", type=int, default=None, help="cfg5: tracklets per video (default 64)")
    ap.add_argument("--roi-streams", type=int, default=0,
                    help="cfg5: HIP streams the RoI head alternates its RoI chunks between (0 = the class default)")
    ap.add_argument("--fused-bottleneck", choices=["auto", "off"], default="auto",
                    help="cfg5: `off` runs every backbone convolution as its own launch (the round-2 path)")
    ap.add_argument("--fused-block", choices=["auto", "off"], default="auto",
                    help="cfg5: `off` runs the identity blocks of res2 / res3 as conv1 + fused tail (round 4) instead of one launch")
    ap.add_argument("--frame-chunk", type=int, default=0,
                    help="cfg5: frames per backbone launch (0 = the class default)")
    ap.add_argument("--tail-io-waves", choices=["auto", "off"], default="auto",
                    help="cfg5: `off` runs res4's fused tails on the one-role kernel (round 3/4) instead of the role-split one (round 5)")
    ap.add_argument("--associate", choices=["off", "process", "thread"], default="off",
                    help="cfg5: frames -> VIDEO-LEVEL relations.  Behind every video's triplets the greedy relational association "
                         "(reference lib/modeling/association.py:117-175; 59 segments x --tracklets tracklets x 200 predictions, "
                         "device IoU tables) runs in a worker process (`thread`: on a thread of this process, the A/B arm) while the GPU works on the next "
                         "video; the line reports the steady-state ms per video with it, the association's time alone and "
                         "beside the GPU pipeline, the GPU's idle fraction and the host's call rate")
    ap.add_argument("--serial-tail", action="store_true",
                    help="A/B: RELPN.OVERLAP_TAIL = False (PPN and decode on the caller's stream, after the encoder)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse only the rank launch + rendezvous on the CPU (gloo), no GPU work")
    ap.add_argument("--stub-gpu", action="store_true",
                    help="CPU rehearsal of the whole rank body (gloo, recorded decoded rows instead of the GPU step)")
    args = ap.parse_args(argv)
    steps, warm = DEFAULT_STEPS[args.workload]
    args.steps = steps if args.steps is None else args.steps
    args.warmup = warm if args.warmup is None else args.warmup
    return args


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """Start one process per GPU (torch.distributed.run) as a child and return its exit code.  Runs before
    torch is imported here: the parent never initialises the GPU, so nothing is re-exec'd after a HIP call."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_weights(sd):
    return {"conv_w": sd[DPN_PRE + "conv.weight"], "conv_b": sd[DPN_PRE + "conv.bias"],
            "dur_w": sd[DPN_PRE + "duration_pred.weight"], "dur_b": sd[DPN_PRE + "duration_pred.bias"],
            "rel_w": sd[DPN_PRE + "relness_pred.weight"], "rel_b": sd[DPN_PRE + "relness_pred.bias"],
            "cls_w": sd["classifier.rel_predictor.weight"], "cls_b": sd["classifier.rel_predictor.bias"]}


def cpu_baseline(weights, runs):
    """Reference-faithful dense forward (oracle.forward_dense: materialise [P,4096,150] -> DPNHead -> heads;
    RelOIPool; predicate head) on a FIXED sample of one cfg2 video: all 31 objects of subjects 0..7
    (248 pairs, 3.75 TFLOP).  One warm-up run, then `runs` timed runs; the median is reported."""
    import numpy as np
    import torch

    import oracle
    import tspn_mi355x as tspn
    cores = usable_cores()
    torch.set_num_threads(cores)
    v = tspn.synth.make_video(1, N_TRK, T_FRAMES, D_ROI)
    feats, boxes = torch.from_numpy(v["tracklet_feats"]), torch.from_numpy(v["tracklet_boxes"])
    w = {k: torch.from_numpy(x) for k, x in weights.items()}
    pairs = oracle.pair_index(N_TRK)[: CPU_SUBJECTS * (N_TRK - 1)]
    p = pairs.shape[0]

    def run():
        t0 = time.perf_counter()
        with torch.no_grad():
            oracle.forward_dense(feats, boxes, pairs, w)
        return time.perf_counter() - t0

    run()  # warm-up (thread pool, oneDNN primitive cache)
    times = sorted(run() for _ in range(max(1, runs)))
    med = float(np.median(times))
    return {"value": p / med, "unit": "tracklet-pairs/s", "cores": cores, "kind": "port",
            "cpu": cpu_model(), "runs": len(times),
            "spread": [p / times[-1], p / times[0]],
            "sample": f"all {N_TRK - 1} objects of {CPU_SUBJECTS} subjects = {p} of 992 pairs of one cfg2 video, dense reference "
                      f"formulation (oracle.forward_dense), median of {len(times)} runs after 1 warm-up "
                      f"({med:.2f} s per run), torch {torch.__version__} CPU, {cores} threads"}


def cpu_baseline_cfg3(weights, n, t, d, pairs_in_sample=96):
    """The oracle's bf16 restatement (oracle.forward_bf16: bf16 operands, the reference's DPNHead / RelationPredictor
    under .bfloat16() pin its rounding points, golden g8) on the first `pairs_in_sample` pairs of one video of the
    shape, ONE run (22.7 GFLOP per cfg3 pair in the dense formulation)."""
    import torch

    import oracle
    import tspn_mi355x as tspn
    cores = usable_cores()
    torch.set_num_threads(cores)
    v = tspn.synth.make_video(1, n, t, d)
    feats = torch.from_numpy(v["tracklet_feats"])
    w = {k: torch.from_numpy(x) for k, x in weights.items()}
    pairs = oracle.pair_index(n)[:pairs_in_sample]
    t0 = time.perf_counter()
    with torch.no_grad():
        oracle.forward_bf16(feats, pairs, w)
    dt = time.perf_counter() - t0
    return {"value": pairs_in_sample / dt, "unit": "tracklet-pairs/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "runs": 1, "seconds": dt,
            "sample": f"the first {pairs_in_sample} of {n * (n - 1)} pairs of one video (N={n}, T={t}, D={d}), bf16-operand "
                      f"dense restatement (oracle.forward_bf16), one run of {dt:.2f} s, torch {torch.__version__} CPU, "
                      f"{cores} threads"}


def cpu_baseline_cfg5(bb_sd, r5_sd, score_w, n, t, h, w_img, frames=8, pairs_in_sample=8):
    """frames -> triplets on the CPU, bounded: `frames` 720p frames through the oracle's ResNet-101-C4 (fp32 torch
    convs), the RoI head of those frames' `n` boxes each (restated ROIAlign + res5), and the bf16 scorer restatement
    on `pairs_in_sample` pairs at T frames; scaled to a video: t/frames * (backbone + RoI head) + n(n-1)/pairs * scorer."""
    import numpy as np
    import torch

    import oracle
    from oracle import roi_head_oracle as ro
    import tspn_mi355x as tspn
    cores = usable_cores()
    torch.set_num_threads(cores)
    img = torch.from_numpy(tspn.hashrng.uniform(9, "img", (frames, h, w_img, 3), -0.5, 0.5))
    bb = {k: torch.from_numpy(v) for k, v in bb_sd.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        fm = ro.resnet_c4(img, bb, tspn.ResNetC4.BLOCKS[101], dtype=torch.float32)
    t_bb = time.perf_counter() - t0
    xy = tspn.hashrng.uniform(9, "xy", (n, frames, 2)) * np.array([900.0, 400.0], np.float32)
    wh = 40 + tspn.hashrng.uniform(9, "wh", (n, frames, 2)) * 260
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], axis=2).astype(np.float32))
    r5 = {k: torch.from_numpy(v) for k, v in r5_sd.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        ro.res5_roi_head(fm, boxes, r5)
    t_roi = time.perf_counter() - t0
    d = 2048
    feats = torch.from_numpy(tspn.hashrng.uniform(9, "f", (4, t, d)))
    wts = {k: torch.from_numpy(x) for k, x in score_w.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        oracle.forward_bf16(feats, oracle.pair_index(4)[:pairs_in_sample], wts)
    t_pairs = time.perf_counter() - t0
    p_vid = n * (n - 1)
    video_s = t / frames * (t_bb + t_roi) + p_vid / pairs_in_sample * t_pairs
    return {"value": p_vid / video_s, "unit": "tracklet-pairs/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "runs": 1, "videos_per_s": 1.0 / video_s,
            "seconds": {"backbone": t_bb, "roi_head": t_roi, "scorer": t_pairs},
            "sample": f"{frames} frames of {h}x{w_img} through the oracle's ResNet-101-C4 ({t_bb:.2f} s), the RoI head of their "
                      f"{n} boxes each ({t_roi:.2f} s), the bf16 scorer restatement on {pairs_in_sample} pairs at T={t}, D={d} "
                      f"({t_pairs:.2f} s); scaled to one video = {t}/{frames} x (backbone + RoI head) + {p_vid}/{pairs_in_sample} x "
                      f"scorer = {video_s:.0f} s; one run each, torch {torch.__version__} CPU, {cores} threads"}


def pmc_traffic(workload, videos, conv):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, produced by tools/pmc_summary.py, one measurement set per
    "<workload>:<videos per launch>"); null if this step size was not measured.  PMC counters cannot be
    read from inside the process."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    kernel = {"winograd6": "conv3_wino63_kernel", "bf16": "conv3_bf16_big_kernel"}.get(conv, "conv3_mfma_cl_kernel")
    try:
        data = json.load(open(path))
        ms = data["sets"][f"{'cfg2' if workload == 'cfg4' else workload}:{videos}"]
        # the profiler names template instances (`conv3_wino63_kernel<true>`): match the instances too, and prefer an entry
        # of the set's own pass over one carried along from an older pass (those keep their own `source`)
        names = [k for k in ms["kernels"] if k == kernel or k.startswith(kernel + "<")]
        name = sorted(names, key=lambda k: ("source" in ms["kernels"][k], k))[0]
        return {"traffic": ms["kernels"][name]["hbm_bytes"],
                "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": ms["kernels"][name].get("source", ms["source"])}
    except (OSError, KeyError, ValueError, TypeError, IndexError):
        return {"traffic": None}


def backbone_traffic(frames, h, w):
    """Fabric bytes of the backbone per video from the committed PMC passes of tools/bench_backbone.py
    (tools/pmc_backbone.py: bytes per 720p frame summed over all backbone launches) x the frames of the video."""
    if (h, w) != (720, 1280):
        return {"traffic": None}
    try:
        ms = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["sets"]["cfg5:1"]
        return {"traffic": ms["backbone_bytes_per_frame"] * frames,
                "traffic_unit": "bytes per video's backbone call (sum over its launches of 2*FETCH_SIZE + WRITE_SIZE, "
                                "measured per 720p frame x frames)", "traffic_source": ms["source"]}
    except (OSError, KeyError, ValueError, TypeError):
        return {"traffic": None}


class ClockSampler:
    """Shader clock (MHz) of the device while the timed steps run, read from the amdgpu hwmon node
    (`freq1_input`, Hz) by a background thread every few ms.  None when the node is not readable."""

    def __init__(self, pci_address):
        """`pci_address` "dddd:bb:dd.f" of the HIP device (the box shows the hwmon nodes of every GPU of
        the host, the process sees one of them)."""
        import glob
        self.path = None
        for node in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            card_dev = node.split("/hwmon/")[0]
            if os.path.basename(os.path.realpath(card_dev)).lower() == pci_address.lower():
                self.path = node
                break
        self.samples = []
        self._stop = False
        self._thread = None

    def _loop(self):
        while not self._stop:
            try:
                self.samples.append(int(open(self.path).read()) / 1e6)
            except (OSError, ValueError):
                return
            time.sleep(0.004)

    def start(self):
        if self.path is not None:
            import threading
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"mean": sum(s) / len(s), "min": s[0], "max": s[-1], "samples": len(s), "source": self.path}


def launch_check(args):
    """CPU rehearsal of the N>1 launch: gloo rendezvous, rank count assertion, one gather, one JSON line."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import tspn_mi355x as tspn
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    out = tspn.dist.gather_results(torch.full((2, 3), float(rank)), 2 * world)
    assert out.shape[0] == 2 * world and out[:, 0].tolist() == [float(r) for r in range(world) for _ in range(2)]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_flag": args.gpus}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ workloads
class Gatherer:
    """The one collective of the path: all-gather of per-video results at the end of a step (RCCL; gloo in the
    CPU rehearsal).  Keeps the last local and gathered rows for the post-run check, and times the collective itself:
    HIP events on the stream it is issued from (RCCL runs on its own stream; torch makes the issuing stream wait for
    it, so the second event fires when the gathered rows are usable), wall clock under gloo."""

    def __init__(self, tspn, args, world, use_dist, rank=0, torch=None, on_gpu=False):
        self.tspn, self.args, self.world, self.use_dist, self.rank = tspn, args, world, use_dist, rank
        self.last, self.gathered = None, None
        self.total = args.total_videos          # None: every rank holds the same number of videos
        self.torch, self.on_gpu = torch, on_gpu
        self.events, self.wall = [], []

    def shard(self, default_b):
        """(first global video, videos of this rank, videos over all ranks) per step."""
        if self.total is None:
            return self.rank * default_b, default_b, self.world * default_b
        lo, hi = self.tspn.dist.shard_range(self.total, self.rank, self.world)
        return lo, hi - lo, self.total

    def __call__(self, sc, trip, tid, idx, logits=None):
        b = sc.shape[0]
        num = self.total if self.total is not None else self.world * b
        if self.use_dist:
            if self.on_gpu:
                e0, e1 = (self.torch.cuda.Event(enable_timing=True) for _ in range(2))
                e0.record()
            else:
                t0 = time.perf_counter()
            if self.args.gather == "decoded" or logits is None:
                self.gathered = self.tspn.dist.gather_decoded(sc, trip, tid, num, pair_proposals=idx, force=True)
            else:
                self.tspn.dist.gather_results(logits, num, force=True)
                self.tspn.dist.gather_results(idx, num, force=True)
            if self.on_gpu:
                e1.record()
                self.events.append((e0, e1))
            else:
                self.wall.append((time.perf_counter() - t0) * 1e3)
        self.last = (sc, trip, tid, idx)

    def gather_ms(self, skip):
        """Mean time of the collective over the timed steps (the first `skip` calls are warm-up); call after a sync."""
        ms = [a.elapsed_time(b) for a, b in self.events[skip:]] if self.on_gpu else self.wall[skip:]
        return float(sum(ms) / len(ms)) if ms else None

    def check(self, rank):
        """Every rank holds every video's decoded rows, in global order; its own block equals what it computed."""
        import torch
        if not self.use_dist or self.gathered is None:
            return
        g, b = self.gathered, self.last[0].shape[0]
        lo, cnt, num = self.shard(b)
        assert cnt == b, (cnt, b)
        assert g["scores"].shape == (num, self.last[0].shape[1]), g["scores"].shape
        assert g["pair_proposals"].shape == (num, self.last[3].shape[1]), g["pair_proposals"].shape
        assert torch.equal(g["scores"][lo:lo + b], self.last[0]), "gathered scores differ from the local block"
        assert torch.equal(g["triplets"][lo:lo + b], self.last[1]), "gathered triplets differ"
        assert torch.equal(g["pair_proposals"][lo:lo + b], self.last[3]), "gathered pair proposals differ"


class StubWorkload:
    """--stub-gpu: recorded decoded rows (deterministic per global video index) in place of the GPU step.
    `--workload cfg5` rehearses that workload's shape: ONE video of 64 tracklets per rank per step."""

    def __init__(self, args, tspn, torch, np, dev, world, rank, gather):
        self.args, self.world, self.rank, self.gather = args, world, rank, gather
        cfg5 = args.workload == "cfg5"
        n_trk = self.n_trk = CFG5[0] if cfg5 else (CFG3[0] if args.workload == "cfg3" else N_TRK)
        first, self.B, self.total_videos = gather.shard(args.videos if args.videos is not None else (1 if cfg5 else 4))
        self.units_per_step = self.B * n_trk * (n_trk - 1)
        self.total_units_per_step = self.total_videos * n_trk * (n_trk - 1)
        rows = []
        for b in range(self.B):
            g = first + b     # global video index
            sc = np.sort(tspn.hashrng.uniform(1000 + g, "sc", (TOPK_SEG,)))[::-1].copy()
            trip = np.stack([tspn.hashrng.integers(1000 + g, "s", (TOPK_SEG,), 0, 35),
                             tspn.hashrng.integers(1000 + g, "p", (TOPK_SEG,), 0, K_PRED),
                             tspn.hashrng.integers(1000 + g, "o", (TOPK_SEG,), 0, 35)], axis=1).astype(np.int64)
            tid = tspn.hashrng.integers(1000 + g, "t", (TOPK_SEG, 2), 0, n_trk).astype(np.int64)
            idx = tspn.hashrng.integers(1000 + g, "i", (TOPK_PPN,), 0, n_trk * n_trk).astype(np.int64)
            rows.append((sc, trip, tid, idx))
        if rows:
            self.rows = tuple(torch.from_numpy(np.stack([r[k] for r in rows])) for k in range(4))
        else:               # an empty shard still takes part in the collective
            self.rows = (torch.zeros((0, TOPK_SEG)), torch.zeros((0, TOPK_SEG, 3), dtype=torch.int64),
                         torch.zeros((0, TOPK_SEG, 2), dtype=torch.int64), torch.zeros((0, TOPK_PPN), dtype=torch.int64))

    def step(self, i):
        self.gather(*self.rows)

    def sync(self):
        pass

    def report(self, elapsed, clock_mhz):
        return {"metric": f"tracklet-pairs/sec scored (N={self.n_trk}, T={T_FRAMES}, D={D_ROI})", "dtype": "none", "stub": True,
                "config": {"workload": f"STUB ({self.args.workload} shape): recorded decoded rows instead of the GPU step (CPU "
                                       "rehearsal of the rank body over gloo; measures nothing)",
                           "videos_per_gpu_per_step": self.B, "videos_per_step": self.total_videos},
                "roofline": None}

    def cpu_baseline(self):
        return None


class ScoringWorkload:
    """cfg2 / cfg4 / cfg3: tracklet tensors -> scores, through BaseModel or at ops level."""

    def __init__(self, args, tspn, torch, np, dev, world, rank, gather):
        self.args, self.tspn, self.torch, self.np, self.dev, self.world, self.rank, self.gather = \
            args, tspn, torch, np, dev, world, rank, gather
        bf16 = self.bf16 = args.workload == "cfg3"
        N, T, D = self.N, self.T, self.D = CFG3 if bf16 else (N_TRK, T_FRAMES, D_ROI)
        C = self.C = 2 * D
        first, B, self.total_videos = gather.shard(args.videos if args.videos is not None else
                                                   {"cfg2": 16, "cfg4": 64, "cfg3": 4}[args.workload])
        self.B = B
        if B < 1:
            raise SystemExit(f"bench.py: rank {rank} got no video of --total-videos {args.total_videos}; give every rank one")
        self.P_vid = N * (N - 1)
        self.P = B * self.P_vid
        self.units_per_step = self.P
        self.total_units_per_step = self.total_videos * self.P_vid
        self.via_model = not args.ops_level
        total_steps = args.warmup + args.steps

        # ---- weights (seed 0) and inputs (seed 1 + global video index), random-init / synthetic
        sd = self.sd = tspn.synth.make_weights(0, c=C, a=A_ANCH, k=K_PRED)
        self.wnp = oracle_weights(sd)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        # `--batches` resident input batches; the first holds hash-RNG videos (seed 1 + global video index, the
        # inputs the parity tests use), the others are drawn on the device from the same U[0,1) distribution
        nb = self.nb = max(1, args.batches)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        hashed = min(B, 16)
        vids = [tspn.synth.make_video(1 + first + b, N, T, D) for b in range(hashed)]
        self.feats_all, self.cls_all, self.boxes_all = [], [], []
        for k in range(nb):
            f = torch.rand((B * N, T, D), device=dev, generator=gen)
            c = torch.rand((B, N, 35), device=dev, generator=gen)
            # integer-valued boxes (l, t, r, b) as in SURVEY.md §8d: x, y in [0, 900), w, h in [10, 300)
            xy = torch.floor(torch.rand((B * N, T, 2), device=dev, generator=gen) * 900.0)
            wh = torch.floor(10.0 + torch.rand((B * N, T, 2), device=dev, generator=gen) * 290.0)
            bx = torch.cat([xy, xy + wh], dim=2).contiguous()
            if k == 0:
                f[: hashed * N] = d(np.concatenate([v["tracklet_feats"] for v in vids]))
                c[:hashed] = d(np.stack([v["track_cls_logits"] for v in vids]))
                bx[: hashed * N] = d(np.concatenate([v["tracklet_boxes"] for v in vids]))
            self.feats_all.append(tspn.ops.cast_bf16(f) if bf16 else f)
            self.cls_all.append(c)
            self.boxes_all.append(bx)
        del vids
        self.events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                       for _ in range(total_steps)]
        self.overlap = args.overlap and not bf16 and not self.via_model
        self.side = torch.cuda.Stream(device=dev) if self.overlap else None
        self.ev_logits = [torch.cuda.Event() for _ in range(total_steps)]
        self.ev_side = [torch.cuda.Event() for _ in range(total_steps)]
        for i, (a, b) in enumerate(self.events):  # create the HIP event handles
            a.record(); b.record(); self.ev_logits[i].record(); self.ev_side[i].record()
        self.geom = None
        self.host = args.host_inputs != "off"
        if self.host:
            if not self.via_model:
                raise SystemExit("bench.py: --host-inputs runs through BaseModel (drop --ops-level)")
            place = (lambda x: x.cpu().pin_memory()) if args.host_inputs.endswith("pinned") else (lambda x: x.cpu())
            self.feats_all = [place(x) for x in self.feats_all]
            self.cls_all = [place(x) for x in self.cls_all]
            self.boxes_all = [place(x) for x in self.boxes_all]
            self.prefetch = None
            if args.host_inputs.startswith("prefetch"):
                def loader():          # what the reference's DataLoader yields (build.py:84-93): host batches, forever
                    i = 0
                    while True:
                        f, c, bx = self.feats_all[i % nb], self.cls_all[i % nb], self.boxes_all[i % nb]
                        yield [tspn.PairList.from_tracklets(f[b * N:(b + 1) * N], bx[b * N:(b + 1) * N], c[b]) for b in range(B)]
                        i += 1
                self.prefetch = iter(tspn.dataset.DevicePrefetcher(loader(), dev))
        if self.via_model:
            self._init_model()
        else:
            self._init_ops(d)
        torch.cuda.synchronize()

    # -- the drop-in surface: BaseModel(cfg) with the synthetic weights, PairLists in, forward + decode
    def _init_model(self):
        tspn, torch = self.tspn, self.torch
        cfg = tspn.load_cfg(None, **{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": self.C,
                                     "PREDICT.FEATURE_DIM": self.C, "RELPN.DPN.NUM_ANCHORS_PER_LOCATION": A_ANCH,
                                     "PREDICT.PREDICATE_NUM": K_PRED, "RELPN.PPN.NUM_PAIR_PROPOSALS": TOPK_PPN,
                                     "RELPN.DPN.PAIR_GEOMETRY": True, "RELPN.OVERLAP_TAIL": not self.args.serial_tail,
                                     "RELPN.DPN.CONV_ALGO": "direct" if self.args.conv == "direct" else "auto",
                                     **({"RELPN.DPN.HOST_CHUNK_VIDEOS": self.args.host_chunk} if self.args.host_chunk else {}),
                                     **({"RELPN.DPN.HOST_CHUNK_VIDEOS": 0} if self.args.host_inputs == "naive" else {})})
        model = tspn.BaseModel(cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in self.sd.items()})
        self.model = model.to(self.dev).eval()

    def _step_model(self, i):
        tspn, torch, N, B = self.tspn, self.torch, self.N, self.B
        feats, cls, boxes = self.feats_all[i % self.nb], self.cls_all[i % self.nb], self.boxes_all[i % self.nb]
        if self.host and self.prefetch is not None:
            plists = next(self.prefetch)
        else:
            plists = [tspn.PairList.from_tracklets(feats[b * N:(b + 1) * N], boxes[b * N:(b + 1) * N], cls[b])
                      for b in range(B)]
        self.model.profile_conv_events(self.events[i])
        pair_props, dur_props, rel_logits = self.model(plists, None)
        dec = self.model.decode(plists, rel_logits, topk_per_pair=TOPK_PAIR, topk_per_seg=TOPK_SEG)
        # the per-video results are slices of the batched tensors the decode / PPN launches wrote: batch them back as a
        # view (torch.stack when they are not laid out that way) -- no copy kernels between the step's last launch and the gather
        def batched(ts):
            v = tspn.model._consecutive_view(ts)
            return torch.stack(ts) if v is None else v.view((len(ts),) + tuple(ts[0].shape))
        sc, trip, tid = (batched([d[k] for d in dec]) for k in range(3))
        self.geom = dur_props[0].geom
        self.gather(sc, trip, tid, batched(list(pair_props)),
                    logits=torch.stack(rel_logits) if self.args.gather == "logits" else None)

    # -- ops level: caller-held packed weights, workspace and outputs
    def _init_ops(self, d):
        tspn, torch, np, dev = self.tspn, self.torch, self.np, self.dev
        args, wnp, B, N, T, D, P = self.args, self.wnp, self.B, self.N, self.T, self.D, self.P
        r16 = lambda x: tspn.ops.cast_bf16(x.contiguous()).float()  # noqa: E731
        conv_w = d(wnp["conv_w"])
        if self.bf16:
            self.packed = tspn.ops.pack_conv3_bf16(conv_w, split=D)
        else:
            self.packed = {"direct": tspn.ops.pack_conv3, "winograd6": tspn.ops.pack_conv3_wino63}[args.conv](conv_w, split=D)
        del conv_w
        self.conv_b = d(wnp["conv_b"])
        self.head_w = d(np.concatenate([wnp["rel_w"][:, :, 0], wnp["dur_w"][:, :, 0]]))
        self.head_b = d(np.concatenate([wnp["rel_b"], wnp["dur_b"]]))
        self.cls_w, self.cls_b = d(wnp["cls_w"]), d(wnp["cls_b"])
        if self.bf16:
            self.conv_b, self.head_b, self.cls_w, self.cls_b = r16(self.conv_b), r16(self.head_b), r16(self.cls_w), r16(self.cls_b)
            self.head_pk = tspn.ops.pack_heads_bf16(self.head_w)
            self.ws = torch.empty(tspn.ops.fused_bf16_workspace_bytes(B, N, T, D, A_ANCH, K_PRED, P), dtype=torch.uint8, device=dev)
        else:
            self.ws = torch.empty(tspn.ops.fused_workspace_bytes(B, N, T, D, A_ANCH, K_PRED, P), dtype=torch.uint8, device=dev)
        self.ppn_w = {k[len(PPN_PRE):]: d(v) for k, v in self.sd.items() if k.startswith(PPN_PRE)}
        self.pairs = torch.cat([tspn.ops.pair_index(N, dev, base=b * N) for b in range(B)]).contiguous()
        self.local_pairs = tspn.ops.pair_index(N, dev).unsqueeze(0).expand(B, -1, -1).contiguous()
        self.out_heads = torch.empty((P, 3 * A_ANCH, T), dtype=torch.float32, device=dev)
        # two logits buffers: the second stream may still be decoding step i - 1 while step i writes its logits
        self.out_logits2 = [torch.empty((P, K_PRED), dtype=torch.float32, device=dev) for _ in range(2)]

    def _tail(self, lg, cls, boxes):
        """What only needs the logits (and the boxes): pair geometry, PPN, top-k decode, result gather."""
        tspn = self.tspn
        # the bbox half of the N^2 pair builder: relative geometry [P, 8, T] of every pair (one lane per
        # (pair, frame), motion channels by wavefront shuffle)
        _, self.geom = tspn.ops.pair_gather(None, boxes, self.pairs, want_feat=False, check_pairs=False)
        _, idx = tspn.ops.ppn_pair_matrix_topk(cls, self.ppn_w, TOPK_PPN)
        # top-k triplet decode (predict.py:66-106): per pair top-20 of 132, per video top-200
        lg3 = lg.view(self.B, self.P_vid, K_PRED)
        sc, trip, tid = tspn.ops.decode_topk(lg3, self.local_pairs, cls, row_mul=1, topk_per_pair=TOPK_PAIR,
                                             topk_per_seg=TOPK_SEG, check_pairs=False)
        self.gather(sc, trip, tid, idx, logits=lg3 if self.args.gather == "logits" else None)

    def _step_ops(self, i):
        tspn, torch = self.tspn, self.torch
        feats, cls, boxes = self.feats_all[i % self.nb], self.cls_all[i % self.nb], self.boxes_all[i % self.nb]
        if self.bf16:
            _, lg = tspn.ops.forward_fused_bf16(feats, self.pairs, self.B, self.N, self.packed, self.conv_b, self.head_pk,
                                                self.head_b, self.cls_w, self.cls_b, workspace=self.ws,
                                                conv_events=self.events[i])
            return self._tail(lg, cls, boxes)
        lg = self.out_logits2[i % 2]
        main = torch.cuda.current_stream()
        if self.overlap and i >= 2:
            main.wait_event(self.ev_side[i - 2])      # the second stream has finished reading this logits buffer
        # the logits are computed first inside the call (they depend on the tracklet means only) and the
        # event fires there: the second stream decodes and gathers under this step's encoder
        tspn.ops.forward_fused(feats, self.pairs, self.B, self.N, self.packed, self.conv_b, self.head_w, self.head_b,
                               self.cls_w, self.cls_b, workspace=self.ws, out_heads=self.out_heads, out_logits=lg,
                               check_pairs=False, conv_events=self.events[i], canonical_pairs=True,
                               logits_event=self.ev_logits[i] if self.overlap else None)
        if not self.overlap:
            return self._tail(lg, cls, boxes)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_logits[i])
            self._tail(lg, cls, boxes)
            self.ev_side[i].record(self.side)

    def step(self, i):
        return self._step_model(i) if self.via_model else self._step_ops(i)

    def sync(self):
        self.torch.cuda.synchronize()

    def report(self, elapsed, clock_mhz):
        np, args, bf16 = self.np, self.args, self.bf16
        N, T, D, C, B, P = self.N, self.T, self.D, self.C, self.B, self.P
        # dominant kernel: the temporal conv of the tracklet projections, HIP events inside the timed steps
        conv_ms = [a.elapsed_time(b) for a, b in self.events[args.warmup:]]
        conv_avg_s = float(np.mean(conv_ms)) * 1e-3
        conv_flop_direct = 2.0 * (2 * C) * (3 * D) * (B * N * T)  # M=2C, K=3D, columns=B*N*T
        # F(6,3) issues 8 channel-GEMMs on a sixth of the columns (ceil(T/6) sextets per tracklet): 4/9 of the direct work
        frac = {"direct": 1.0, "winograd6": (4.0 / 9.0) * (6 * -(-T // 6)) / T}[args.conv]
        conv_flop = conv_flop_direct * (1.0 if bf16 else frac)
        if self.host and self.prefetch is None and getattr(self.model, "host_chunk_videos", 0) > 0:
            # host-resident inputs are scored chunk by chunk (model._HostPipeline): the events bracket the LAST chunk's launch
            chunks = type(self.model)._host_chunk_schedule(B, self.model.host_chunk_videos)
            share = (chunks[-1][1] - chunks[-1][0]) / float(B)
            conv_flop_direct, conv_flop = conv_flop_direct * share, conv_flop * share
        achieved = conv_flop / conv_avg_s / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        cfg_name = {"cfg2": "BASELINE cfg2: synthetic VidVRD shape N=32 T=150 D=2048 (C=4096, A=4, K=132), fp32, "
                            "random-init weights",
                    "cfg4": "BASELINE cfg4 shard: 64 of the 512 synthetic VidVRD-shaped videos per GPU per step "
                            "(N=32 T=150 D=2048, C=4096, A=4, K=132), fp32, random-init weights",
                    "cfg3": "BASELINE cfg3: VidOR long-clip shape N=64 T=900 D=1024 (C=2048, A=4, K=132), "
                            "bf16 operands / fp32 accumulation, random-init weights"}[args.workload]
        gather_txt = ""
        if self.gather.use_dist:
            gather_txt = (" + RCCL all-gather of the decoded results (10.8 KB per video)" if args.gather == "decoded"
                          else " + RCCL all-gather of logits/top-k")
        return {
            "metric": f"tracklet-pairs/sec scored (N={N}, T={T}, D={D})",
            "dtype": "bf16" if bf16 else "f32",
            "config": {"workload": cfg_name,
                       "videos_per_gpu_per_step": B, "pairs_per_video": self.P_vid, "resident_input_batches": self.nb,
                       "surface": (("BaseModel.forward(pair_list) + BaseModel.decode on HOST-resident PairLists ("
                                    + args.host_inputs + " memory), host results: PCIe inclusive") if self.host else
                                   "BaseModel.forward(pair_list) + BaseModel.decode on device-resident PairLists"
                                   if self.via_model else "ops.forward_fused (pre-allocated workspace and outputs)"),
                       "host_inputs": args.host_inputs,
                       "path": ("fused/factorised (tspn_forward_fused_bf16)" if bf16 else
                                "fused/factorised (tspn_forward_fused_f32)")
                               + " + pair geometry + PPN top-k + top-k triplet decode" + gather_txt
                               + (" (geometry, PPN, decode and gather on a second stream under the encoder)" if self.overlap
                                  else ""),
                       "dense_equivalent_gflop_per_pair": (2.0 * T * C * (3 * C + 3 * A_ANCH) + 2.0 * C * K_PRED) / 1e9,
                       "conv_algo": "direct" if bf16 else args.conv,
                       "executed_gflop_per_pair": (conv_flop + 2.0 * P * T * C * 3 * A_ANCH + 2.0 * P * C * K_PRED) / P / 1e9},
            "roofline": {"bound": "mfma",
                         "kernel": ("conv3_bf16_big_kernel (tracklet projections: k=3 conv as bf16 32x32x16 MFMA "
                                    "implicit GEMM, M=2C, K=3D)" if bf16 else
                                    "conv3_wino63_kernel (tracklet projections: k=3 conv, Winograd F(6,3), fp32 32x32x2 "
                                    "MFMA, M=2C, 8 channel-GEMMs of K=D on a sixth of the columns)"
                                    if args.conv == "winograd6" else
                                    "conv3_mfma_cl_kernel (tracklet projections: k=3 conv as fp32 32x32x2 "
                                    "MFMA implicit GEMM, M=2C, K=3D)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         **pmc_traffic(args.workload, B, "bf16" if bf16 else args.conv),
                         "direct_equivalent_tflops": conv_flop_direct / conv_avg_s / 1e12,
                         "flop_per_launch": conv_flop, "avg_launch_ms": conv_avg_s * 1e3,
                         "min_launch_ms": float(np.min(conv_ms)), "max_launch_ms": float(np.max(conv_ms)),
                         "launch_ms": [round(float(v), 2) for v in conv_ms],
                         "share_of_step": conv_avg_s / (elapsed / args.steps),
                         "clock_mhz": clock_mhz,
                         "frac_at_held_clock": (achieved / (peak * clock_mhz["mean"] / PEAK_CLOCK_MHZ)
                                                if clock_mhz else None)},
        }

    def cpu_baseline(self):
        if self.bf16:
            return cpu_baseline_cfg3(self.wnp, self.N, self.T, self.D)
        return cpu_baseline(self.wnp, self.args.cpu_runs)


def backbone_conv_flops(tspn, h, w, depth=101):
    """2 * MACs of every convolution of the C4 backbone for one frame (real channel counts: the stem's zero
    channel and the halo recomputation of fused bottlenecks are executed but not counted)."""
    def out(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = out(h, 7, 2, 3), out(w, 7, 2, 3)
    fl = 2.0 * h * w * 64 * 3 * 49
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    cin, cout = 64, 256
    for i, nb in enumerate(tspn.ResNetC4.BLOCKS[depth]):
        for b in range(nb):
            s = 2 if (b == 0 and i > 0) else 1
            mid = cout // 4
            h2, w2 = out(h, 1, s, 0), out(w, 1, s, 0)
            fl += 2.0 * h2 * w2 * (mid * cin + 9 * mid * mid + cout * mid + (cout * cin if cin != cout else 0))
            h, w, cin = h2, w2, cout
        cout *= 2
    return fl


class Cfg5Workload:
    """BASELINE configs[4]: frames -> triplets at VidOR scale.  One video per GPU per step: T frames of 720p resident in
    HBM (fp32, mean-subtracted) -> ResNetC4 (R-101, bf16) -> res4 maps -> Res5RoIHead over N x T boxes ->
    tracklet_feats bf16 [N,T,2048] -> BaseModel.forward (bf16 scorer, PPN) -> BaseModel.decode."""

    def __init__(self, args, tspn, torch, np, dev, world, rank, gather):
        self.args, self.tspn, self.torch, self.np, self.dev, self.world, self.rank, self.gather = \
            args, tspn, torch, np, dev, world, rank, gather
        N, T, H, W = CFG5
        N = self.N = args.tracklets or N
        T = self.T = args.frames or T
        self.H, self.W = H, W
        _, self.B, self.total_videos = gather.shard(args.videos if args.videos is not None else 1)
        if self.B != 1:
            raise SystemExit("bench.py --workload cfg5 scores one video per GPU per step (--videos 1, --total-videos = --gpus)")
        D = self.D = 2048
        self.P_vid = N * (N - 1)
        self.units_per_step = self.P_vid
        self.total_units_per_step = self.total_videos * self.P_vid
        t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}  # noqa: E731
        self.bb_sd = tspn.synth.make_backbone_weights(0)
        self.r5_sd = tspn.synth.make_res5_weights(0)
        self.sd = tspn.synth.make_weights(0, c=2 * D, a=A_ANCH, k=K_PRED)
        # frames per backbone launch: ResNetC4.frame_chunk (90 since round 6, see there); x 2 streams (ResNetC4.streams)
        self.net = tspn.ResNetC4(depth=101) if args.frame_chunk <= 0 else tspn.ResNetC4(depth=101, frame_chunk=args.frame_chunk)
        self.net.load_state_dict(t(self.bb_sd))
        self.net = self.net.to(dev)
        if hasattr(self.net, "fuse_bottlenecks"):
            self.net.fuse_bottlenecks = args.fused_bottleneck == "auto"
        if hasattr(self.net, "fuse_blocks"):
            self.net.fuse_blocks = args.fused_block == "auto"
        if hasattr(self.net, "tail_io_waves"):
            self.net.tail_io_waves = args.tail_io_waves == "auto"
        self.head = tspn.Res5RoIHead()
        self.head.load_state_dict(t(self.r5_sd))
        self.head = self.head.to(dev)
        if getattr(args, "roi_streams", 0) > 0:
            self.head.streams = args.roi_streams
        cfg = tspn.load_cfg(None, **{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                     "PREDICT.FEATURE_DIM": 2 * D, "RELPN.DPN.NUM_ANCHORS_PER_LOCATION": A_ANCH,
                                     "PREDICT.PREDICATE_NUM": K_PRED, "RELPN.PPN.NUM_PAIR_PROPOSALS": TOPK_PPN})
        self.model = tspn.BaseModel(cfg)
        self.model.load_state_dict(t(self.sd))
        self.model = self.model.to(dev).eval()
        nb = self.nb = max(1, min(args.batches, 2))
        gen = torch.Generator(device=dev).manual_seed(4321 + rank)
        self.imgs, self.boxes, self.cls = [], [], []
        for _ in range(nb):
            self.imgs.append(torch.rand((T, H, W, 3), device=dev, generator=gen) - 0.5)
            xy = torch.rand((N, T, 2), device=dev, generator=gen) * torch.tensor([900.0, 400.0], device=dev)
            wh = 40 + torch.rand((N, T, 2), device=dev, generator=gen) * 260
            self.boxes.append(torch.cat([xy, xy + wh], dim=2).contiguous())
            self.cls.append(torch.rand((N, 35), device=dev, generator=gen))
        total = args.warmup + args.steps
        self.events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(total)]
        for ev in self.events:
            for e in ev:
                e.record()
        torch.cuda.synchronize()
        self.assoc = None
        if getattr(args, "associate", "off") != "off":
            self._init_association()

    # -- frames -> video-level relations: the association of video k on a worker thread beside video k + 1 on the GPU
    def _init_association(self):
        import concurrent.futures
        tspn = self.tspn
        gold = os.path.join(ROOT, "tests", "golden")
        if gold not in sys.path:
            sys.path.insert(0, gold)
        import cases     # the synthetic multi-segment video of the association tests (a generator, not the oracle)
        nseg = max(2, (self.T - 30) // 15 + 1)                 # 30-frame segments, stride 15 (lib/modeling/__init__.py:35-41)
        rels, trajs = cases.g9_scenario(seed=31, n_seg=nseg, n_trk=self.N, n_pred=TOPK_SEG)
        # the association runs in a worker PROCESS (tspn.association.AssociationWorker): on a thread of this process its
        # ~0.3 s of pure Python per video holds the interpreter lock while this thread issues ~2 000 launches per video
        # (489 ms per video against 385 without it, profiles/r6/cfg5_associate.md).  A hand-over thread waits for the
        # video's triplets (an event), ships the short-term relations to the worker and collects its answer -- all of it
        # blocking calls that do not hold the lock.
        if self.args.associate == "process":
            worker = tspn.association.AssociationWorker(device=str(self.dev), max_traj_num_in_clip=TOPK_SEG)
            self._assoc_worker = worker

            # what the decode of a video's segments yields: arrays per segment (scores [K], triplets [K,3], pairs [K,2])
            # and the tracklet boxes [N,30,4]
            np = self.np
            segs = [index for index, _ in rels]
            arr = {"scores": [np.array([p[0] for p in pr[0]]) for _, pr in rels],
                   "triplets": [np.stack([p[1] for p in pr[0]]) for _, pr in rels],
                   "pairs": [np.stack([p[2] for p in pr[0]]) for _, pr in rels],
                   "boxes": [trajs[index] for index in segs]}

            def run(wait_event):
                if wait_event is not None:
                    wait_event.synchronize()                    # this video's triplets exist (blocks this thread only)
                worker.submit_arrays("video", segs, arr["scores"], arr["triplets"], arr["pairs"], arr["boxes"],
                                     return_relations=False)
                _, count, ms, _ = worker.result()
                return ms, count
        else:                                                   # A/B arm: the same work on a thread of THIS process, own stream
            import copy
            torch = self.torch
            stream = torch.cuda.Stream(device=self.dev)

            def run(wait_event):
                r = copy.deepcopy(rels)                         # the association sorts / aliases its input in place
                if wait_event is not None:
                    wait_event.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(stream):                 # torch's current stream is per thread
                    out = tspn.association.greedy_relational_association(None, r, max_traj_num_in_clip=TOPK_SEG,
                                                                         trajectories=trajs, device=self.dev)
                return (time.perf_counter() - t0) * 1e3, len(out)

        run(None)                                               # warm-up
        alone = sorted(run(None)[0] for _ in range(3))
        self.assoc = {"pool": concurrent.futures.ThreadPoolExecutor(1), "run": run, "pending": None, "alone_ms": alone[1],
                      "beside_ms": [], "wait_ms": [], "issue_ms": [], "relations": None, "segments": nseg, "calls": 0}
        # host call rate: every ABI call of the library passes _abi.check
        check = tspn._abi.check
        st = self.assoc

        def counting_check(rc):
            st["calls"] += 1
            return check(rc)
        tspn._abi.check = counting_check

    def step(self, i):
        tspn, torch = self.tspn, self.torch
        img, boxes, cls = self.imgs[i % self.nb], self.boxes[i % self.nb], self.cls[i % self.nb]
        e0, e1, e2, e3 = self.events[i]
        t_issue = time.perf_counter()
        e0.record()
        maps = self.net(img, bf16=True)                 # [T, H/16, W/16, 1024] bf16
        e1.record()
        feats = self.head(maps, boxes)                  # [N, T, 2048] bf16
        e2.record()
        plist = tspn.PairList.from_tracklets(feats, boxes, cls)
        pair_props, _, rel_logits = self.model([plist], None)
        dec = self.model.decode([plist], rel_logits, topk_per_pair=TOPK_PAIR, topk_per_seg=TOPK_SEG)
        e3.record()
        if self.assoc is not None:
            st = self.assoc
            st["issue_ms"].append((time.perf_counter() - t_issue) * 1e3)
            if st["pending"] is not None:                       # one video in flight on the worker: a pipeline of depth one
                t0 = time.perf_counter()
                ms, st["relations"] = st["pending"].result()
                st["wait_ms"].append((time.perf_counter() - t0) * 1e3)
                st["beside_ms"].append(ms)
            st["pending"] = st["pool"].submit(st["run"], e3)
        sc, trip, tid = (torch.stack([d[k] for d in dec]) for k in range(3))
        self.gather(sc, trip, tid, torch.stack(pair_props))

    def sync(self):
        self.torch.cuda.synchronize()
        if self.assoc is not None and self.assoc["pending"] is not None:   # the last video's association belongs to the job:
            st = self.assoc                                                  # the pipeline drains (reported as drain_ms)
            t0 = time.perf_counter()
            ms, st["relations"] = st["pending"].result()
            st["drain_ms"] = (time.perf_counter() - t0) * 1e3
            st["beside_ms"].append(ms)
            st["pending"] = None

    def _association_report(self, elapsed):
        np, args, st = self.np, self.args, self.assoc
        k = args.steps
        span = [e[0].elapsed_time(e[3]) for e in self.events[args.warmup:]]
        issue = st["issue_ms"][-k:]
        calls = st["calls"] / float(len(st["issue_ms"]) or 1)
        return {"what": "greedy relational association of video k (reference lib/modeling/association.py:117-175) "
                        + ("in a worker PROCESS (tspn.association.AssociationWorker, its own HIP context for the IoU tables)"
                           if args.associate == "process" else "on a THREAD of the driving process (own HIP stream)")
                        + " while the GPU works on video k + 1; one video in flight on the worker",
                "mode": args.associate,
                "segments": st["segments"], "predictions_per_segment": TOPK_SEG, "relations_per_video": st["relations"],
                "ms_per_video_with_association": elapsed / k * 1e3,
                "drain_ms": st.get("drain_ms", 0.0),
                "steady_state_ms_per_video": (elapsed * 1e3 - st.get("drain_ms", 0.0)) / k,
                "note": "ms_per_video_with_association = wall time of the K timed videos INCLUDING the drain of the last video's "
                        "association behind the GPU's last kernel (drain_ms, paid once per run, not per video); "
                        "steady_state_ms_per_video excludes it",
                "association_ms_alone": st["alone_ms"], "association_ms_beside_the_gpu_pipeline": float(np.mean(st["beside_ms"][-k:])),
                "main_thread_wait_for_worker_ms": float(np.mean(st["wait_ms"][-k:])),
                "host_issue_ms_per_video": float(np.mean(issue)), "abi_calls_per_video": calls,
                "abi_calls_per_s_while_issuing": calls / (float(np.mean(issue)) * 1e-3),
                "gpu_ms_per_video": float(np.mean(span)),
                "gpu_idle_fraction": max(0.0, 1.0 - float(np.sum(span)) * 1e-3 / elapsed)}

    def report(self, elapsed, clock_mhz):
        np, args, tspn = self.np, self.args, self.tspn
        ev = self.events[args.warmup:]
        bb_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        roi_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        sc_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
        flop = backbone_conv_flops(tspn, self.H, self.W) * self.T
        achieved = flop / (bb_ms * 1e-3) / 1e12
        return {
            "metric": f"tracklet-pairs/sec scored (N={self.N}, T={self.T}, D={self.D})",
            "dtype": "bf16",
            "config": {"workload": f"BASELINE cfg5: frames -> triplets at VidOR scale: {self.T} frames of {self.H}x{self.W} "
                                   f"and {self.N} tracklets per video, ResNet-101-C4 backbone + Res5 RoI head (bf16 MFMA convs, "
                                   "the stem included) -> tracklet_feats [N,T,2048] bf16 -> BaseModel.forward (bf16 scorer) + "
                                   "BaseModel.decode; random-init weights, one video per GPU per step",
                       "videos_per_gpu_per_step": 1, "pairs_per_video": self.P_vid, "resident_input_batches": self.nb,
                       "videos_per_s": self.world * args.steps / elapsed,
                       "frames_per_s": self.world * args.steps * self.T / elapsed,
                       "rois_per_s": self.world * args.steps * self.T * self.N / elapsed,
                       "stage_ms": {"backbone": bb_ms, "roi_head": roi_ms, "scoring_and_decode": sc_ms},
                       "backbone_ms_per_frame": bb_ms / self.T,
                       "fused_bottlenecks": bool(getattr(self.net, "fuse_bottlenecks", False)),
                       "fused_blocks": bool(getattr(self.net, "fuse_bottlenecks", False) and getattr(self.net, "fuse_blocks", False)),
                       "tail_io_waves": bool(getattr(self.net, "fuse_bottlenecks", False) and getattr(self.net, "tail_io_waves", False)),
                       **({"association": self._association_report(elapsed)} if self.assoc is not None else {})},
            "roofline": {"bound": "mfma",
                         "kernel": "ResNet-101-C4 backbone, all convolutions of one video (bf16 32x32x16 MFMA implicit GEMMs: "
                                   "tail_io_bf16_kernel / bottleneck_block_bf16_kernel / conv2d_nhwc_bf16_kernel / stem_conv_bf16_kernel)",
                         "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_MFMA_TFLOPS, **backbone_traffic(self.T, self.H, self.W),
                         "flop_per_launch": flop, "avg_launch_ms": bb_ms,
                         "flop_note": "algorithmic conv FLOP of the backbone per video (251 GFLOP per 720p frame; halo "
                                      "recomputation and channel padding executed but not counted) / HIP-event time of the "
                                      "backbone call inside the timed steps",
                         "share_of_step": bb_ms * 1e-3 / (elapsed / args.steps),
                         "clock_mhz": clock_mhz},
        }

    def cpu_baseline(self):
        return cpu_baseline_cfg5(self.bb_sd, self.r5_sd, oracle_weights(self.sd), self.N, self.T, self.H, self.W)


def secondary_legs(args, tspn, torch, np, dev):
    """Short runs of the OTHER workloads behind the timed headline region (never inside it), in the same process, so that
    the driver's one bench line also carries driver-observed numbers for them: the reference-order fp32 conv
    (`--conv direct`), one GPU's 64-video shard of cfg4, cfg3 (bf16 long clips) and cfg5 (frames -> triplets).  Each leg is what `python bench.py --workload X`
    times, with fewer steps; a leg that fails reports its error instead of taking the headline line down."""
    import copy
    import gc
    legs = {}
    for name, over in (("cfg2_direct", {"workload": "cfg2", "conv": "direct", "steps": 3, "warmup": 1}),
                       ("cfg4_shard", {"workload": "cfg4", "steps": 3, "warmup": 1}),
                       ("cfg3", {"workload": "cfg3", "steps": 5, "warmup": 2}),
                       ("cfg5", {"workload": "cfg5", "steps": 2, "warmup": 1})):
        a = copy.copy(args)
        a.videos = a.total_videos = None
        for k, v in over.items():
            setattr(a, k, v)
        t_leg = time.perf_counter()
        wl = None
        try:
            g = Gatherer(tspn, a, 1, False, rank=0, torch=torch, on_gpu=True)
            wl = (Cfg5Workload if a.workload == "cfg5" else ScoringWorkload)(a, tspn, torch, np, dev, 1, 0, g)
            for i in range(a.warmup):
                wl.step(i)
            wl.sync()
            t0 = time.perf_counter()
            for i in range(a.warmup, a.warmup + a.steps):
                wl.step(i)
            wl.sync()
            el = time.perf_counter() - t0
            rep = wl.report(el, None)
            leg = {"metric": rep["metric"], "value": wl.total_units_per_step * a.steps / el, "unit": "tracklet-pairs/s",
                   "ms_per_step": el / a.steps * 1e3, "steps": a.steps, "warmup": a.warmup, "dtype": rep["dtype"],
                   "workload": rep["config"]["workload"],
                   "roofline": {k: rep["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                                    "traffic", "traffic_source")}}
            for k in ("videos_per_gpu_per_step", "conv_algo", "stage_ms", "backbone_ms_per_frame", "videos_per_s", "frames_per_s"):
                if k in rep["config"]:
                    leg[k] = rep["config"][k]
        except Exception as exc:   # noqa: BLE001 -- the headline line must still be printed
            leg = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        leg["wall_s"] = round(time.perf_counter() - t_leg, 2)
        legs[name] = leg
        del wl
        gc.collect()
        torch.cuda.empty_cache()
    return legs


# ------------------------------------------------------------------------------------------------ rank body
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.launch_check:
        return launch_check(args)

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist

    import tspn_mi355x as tspn

    torch.set_num_threads(usable_cores())   # torch sizes its pool by the visible cores; the box grants a quota of them
    # stdout carries exactly ONE line, the JSON result: native libraries write there too (RCCL prints a
    # version banner to stdout when the first communicator is created), so fd 1 points at stderr until
    # the result is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); every rank is one "
                         "GPU, so the two must agree")
    use_dist = world > 1 or args.force_collective
    stub = args.stub_gpu
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if torch.cuda.device_count() < local_world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} starts {local_world} rank(s) on this node, one GPU each, but only "
                             f"{torch.cuda.device_count()} HIP device(s) are visible (check HIP_VISIBLE_DEVICES / "
                             "ROCR_VISIBLE_DEVICES)")
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus

    gather = Gatherer(tspn, args, world, use_dist, rank=rank, torch=torch, on_gpu=not stub)
    wl_cls = StubWorkload if stub else (Cfg5Workload if args.workload == "cfg5" else ScoringWorkload)
    wl = wl_cls(args, tspn, torch, np, dev, world, rank, gather)
    total_steps = args.warmup + args.steps

    for i in range(args.warmup):
        wl.step(i)
    if use_dist:
        dist.barrier()
    wl.sync()
    clock = None
    if not stub:
        props = torch.cuda.get_device_properties(dev)
        clock = ClockSampler("%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                   getattr(props, "pci_device_id", 0)))
        clock.start()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        wl.step(i)
    if use_dist:
        dist.barrier()
    wl.sync()
    elapsed = time.perf_counter() - t0
    clock_mhz = clock.stop() if clock is not None else None
    per_rank = None
    if use_dist:
        # every rank's own elapsed time and its mean time inside the collective travel to rank 0 with the maximum: the
        # first multi-GPU run then says by itself whether a shortfall is a slow rank, the gather, or the hosts' cores
        gms = gather.gather_ms(args.warmup)
        mine = torch.tensor([elapsed, gms if gms is not None else -1.0, float(wl.units_per_step)], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)   # the slowest rank sets the job's time
        elapsed = float(tmax.item())
        rows = [[float(v) for v in r.tolist()] for r in allr]
        ms = [r[0] / args.steps * 1e3 for r in rows]
        per_rank = {"ms_per_step": [round(v, 4) for v in ms], "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
                    "gather_ms": [round(r[1], 4) for r in rows], "gather_ms_max": max(r[1] for r in rows),
                    "units_per_step": [int(r[2]) for r in rows],
                    "note": "gather_ms = the all-gather of the decoded rows alone (HIP events on the issuing stream; wall "
                            "clock under gloo), inside ms_per_step; a rank that waits for a slower one shows it here"}
        gather.check(rank)
        per_rank["gathered_equals_local"] = True     # Gatherer.check raised otherwise: every rank holds every video's rows
        per_rank["backend"] = dist.get_backend()

    if rank == 0:
        rep = wl.report(elapsed, clock_mhz)
        out = {
            "metric": rep["metric"],
            "value": wl.total_units_per_step * args.steps / elapsed,
            "unit": "tracklet-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": rep["dtype"],
            "data": "synthetic" if args.host_inputs == "off" else f"synthetic, {args.host_inputs} HOST memory (PCIe inclusive)",
            "config": rep["config"],
            "roofline": rep["roofline"],
        }
        if rep.get("stub"):
            out["stub"] = True
        if per_rank is not None:
            out["per_rank"] = per_rank
        if (world == 1 and not use_dist and not stub and not args.no_secondary and args.workload == "cfg2"
                and args.conv == "winograd6" and not args.ops_level and args.host_inputs == "off"):
            out["secondary"] = secondary_legs(args, tspn, torch, np, dev)
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = wl.cpu_baseline()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
