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

The step runs THROUGH THE DROP-IN SURFACE by default: B device-resident `PairList.from_tracklets` per step ->
`BaseModel.forward(pair_list)` (reference lib/modeling/model.py:53-65, called as predict.py:57 calls it) ->
`BaseModel.decode` (predict.py:59-117).  `--ops-level` times the same kernels through `ops.forward_fused` with
caller-held workspace and outputs instead (the two agree within 1 %, profiles/r3/bench_via_model_vs_ops.md).

A "step" = one pass of the hot path over one batch of synthetic videos per GPU (inputs resident in HBM, a
rotation of `--batches` different batches): tracklet tensors -> [pair builder + temporal encoder +
relationness/span heads + RelOIPool + predicate head] (tspn_forward_fused_f32) + pair geometry [P,8,T] from the
boxes (tspn_pair_gather_f32) + PPN pair-matrix/top-k + top-k triplet decode (tspn_decode_topk_f32, the reference's predict.py:66-106).  The predicate logits are ready
before the encoder starts (they depend on the tracklet means only; tspn_fused_desc.ev_logits_ready), so with
`--overlap` geometry, PPN, decode and the result gather run on a second HIP stream under the encoder of the same
step — on one GPU that buys nothing (the encoder fills the chip), so it is off by default.
Videos shard across ranks
(weak scaling, no collective in the forward); with N>1 each step ends with ONE RCCL all-gather of the
DECODED per-video results — top-200 (score, triplet, pair) + top-256 pair proposals, 10.8 KB per video
(`--gather logits` gathers the 524 KB of predicate logits per video instead, as round 1 did).

Printed JSON (rank 0): see the task contract; extras:
  roofline     dominant kernel = the temporal conv of the tracklet projections (fp32 MFMA; Winograd
               F(6,3) by default, --conv direct for the direct taps);
               achieved = executed FLOP per launch / HIP-event time of that launch inside the
               timed steps (events recorded on the launch stream by the C ABI's hook);
               clock_mhz = shader clock sampled from the driver while the timed steps ran (box-to-box
               variance of `frac` is mostly the clock the chip holds under this kernel).
  cpu_baseline the oracle's reference-faithful dense forward on a FIXED sample of pairs (all 31 objects
               of 8 subjects = 248 pairs), median of 5 runs after a warm-up, on the cores this process may use.
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
PEAK_CLOCK_MHZ = 2400.0  # the clock the 157.3 TFLOP/s figure is quoted at
# --workload cfg3: BASELINE.json configs[2] (VidOR long-clip shape, bf16 operands); not the headline
CFG3 = (64, 900, 1024)
DPN_PRE = "relpn.duration_proposal_network.dpn_head."
PPN_PRE = "relpn.pair_proposal_network.ppn_head."
CPU_SUBJECTS = 8   # cpu_baseline sample: all 31 objects of this many subjects
TOPK_PAIR, TOPK_SEG, TOPK_PPN = 20, 200, 256   # PREDICT.TOPK_PER_PAIR / TOPK_PER_SEG, PPN.NUM_PAIR_PROPOSALS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--videos", type=int, default=None,
                    help="videos per GPU per step (default: 16 for cfg2, 64 for cfg4, 4 for cfg3)")
    ap.add_argument("--batches", type=int, default=2,
                    help="different input batches resident in HBM, rotated step by step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed runs of the CPU baseline (median reported)")
    ap.add_argument("--conv", choices=["winograd6", "direct"], default="winograd6",
                    help="temporal-conv algorithm of the tracklet projections (both fp32 MFMA): Winograd F(6,3) "
                         "(default, what RELPN.DPN.CONV_ALGO = auto selects at this shape) or the direct taps")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4"], default="cfg2",
                    help="cfg2 = headline (N=32,T=150,D=2048, fp32, 16 videos/GPU/step); cfg4 = its 64-videos-per-GPU "
                         "shard of the 512-video batch; cfg3 = N=64,T=900,D=1024 bf16 operands")
    ap.add_argument("--gather", choices=["decoded", "logits"], default="decoded",
                    help="payload of the N>1 result gather: decoded top-k results (default) or raw logits")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the RCCL result gather even with one rank (exercises the N>1 code path)")
    ap.add_argument("--overlap", action="store_true",
                    help="pair geometry, PPN, decode and the result gather on a second HIP stream behind the logits-ready "
                         "event, under the encoder of the same step (measured on one GPU: 32.70 vs 32.65 ms per step, the "
                         "encoder leaves no idle units to fill, so the default keeps one stream)")
    ap.add_argument("--ops-level", action="store_true",
                    help="time ops.forward_fused with pre-allocated workspace / outputs instead of BaseModel.forward + "
                         "BaseModel.decode on PairLists (the default)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse only the rank launch + rendezvous on the CPU (gloo), no GPU work")
    return ap.parse_args()


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
        return {"traffic": ms["kernels"][kernel]["hbm_bytes"],
                "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": ms["kernels"][kernel].get("source", ms["source"])}
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
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus
    bf16 = args.workload == "cfg3"
    N, T, D = CFG3 if bf16 else (N_TRK, T_FRAMES, D_ROI)
    C = 2 * D
    B = args.videos if args.videos is not None else {"cfg2": 16, "cfg4": 64, "cfg3": 4}[args.workload]
    P_vid = N * (N - 1)

    # ---- weights (seed 0) and inputs (seed 1 + global video index), random-init / synthetic
    sd = tspn.synth.make_weights(0, c=C, a=A_ANCH, k=K_PRED)
    wnp = {"conv_w": sd[DPN_PRE + "conv.weight"], "conv_b": sd[DPN_PRE + "conv.bias"],
           "dur_w": sd[DPN_PRE + "duration_pred.weight"], "dur_b": sd[DPN_PRE + "duration_pred.bias"],
           "rel_w": sd[DPN_PRE + "relness_pred.weight"], "rel_b": sd[DPN_PRE + "relness_pred.bias"],
           "cls_w": sd["classifier.rel_predictor.weight"], "cls_b": sd["classifier.rel_predictor.bias"]}
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    via_model = not args.ops_level
    conv_w = d(wnp["conv_w"])
    r16 = lambda x: tspn.ops.cast_bf16(x.contiguous()).float()  # noqa: E731
    if via_model:
        packed = None          # BaseModel packs (and caches) its own device copies
    elif bf16:
        packed = tspn.ops.pack_conv3_bf16(conv_w, split=D)
    else:
        packed = {"direct": tspn.ops.pack_conv3, "winograd6": tspn.ops.pack_conv3_wino63}[args.conv](conv_w, split=D)
    del conv_w
    conv_b = d(wnp["conv_b"])
    head_w = d(np.concatenate([wnp["rel_w"][:, :, 0], wnp["dur_w"][:, :, 0]]))
    head_b = d(np.concatenate([wnp["rel_b"], wnp["dur_b"]]))
    cls_w, cls_b = d(wnp["cls_w"]), d(wnp["cls_b"])
    if bf16:
        conv_b, head_b, cls_w, cls_b = r16(conv_b), r16(head_b), r16(cls_w), r16(cls_b)
        head_pk = tspn.ops.pack_heads_bf16(head_w)
    ppn_w = {k[len(PPN_PRE):]: d(v) for k, v in sd.items() if k.startswith(PPN_PRE)}

    # `--batches` resident input batches; the first holds hash-RNG videos (seed 1 + global video index, the
    # inputs the parity tests use), the others are drawn on the device from the same U[0,1) distribution
    nb = max(1, args.batches)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    hashed = min(B, 16)
    vids = [tspn.synth.make_video(1 + rank * B + b, N, T, D) for b in range(hashed)]
    feats_all, cls_all, boxes_all = [], [], []
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
        feats_all.append(tspn.ops.cast_bf16(f) if bf16 else f)
        cls_all.append(c)
        boxes_all.append(bx)
    del vids
    pairs = torch.cat([tspn.ops.pair_index(N, dev, base=b * N) for b in range(B)]).contiguous()
    local_pairs = tspn.ops.pair_index(N, dev).unsqueeze(0).expand(B, -1, -1).contiguous()
    P = pairs.shape[0]

    ws = None if (bf16 or via_model) else torch.empty(tspn.ops.fused_workspace_bytes(B, N, T, D, A_ANCH, K_PRED, P),
                                                      dtype=torch.uint8, device=dev)
    out_heads = None if via_model else torch.empty((P, 3 * A_ANCH, T), dtype=torch.float32, device=dev)
    # two logits buffers: the second stream may still be decoding step i - 1 while step i writes its logits
    out_logits2 = None if via_model else [torch.empty((P, K_PRED), dtype=torch.float32, device=dev) for _ in range(2)]
    state = {}
    total_steps = args.warmup + args.steps
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(total_steps)]
    overlap = args.overlap and not bf16 and not via_model
    side = torch.cuda.Stream(device=dev) if overlap else None
    ev_logits = [torch.cuda.Event() for _ in range(total_steps)]
    ev_side = [torch.cuda.Event() for _ in range(total_steps)]
    for i, (a, b) in enumerate(events):  # create the HIP event handles
        a.record(); b.record(); ev_logits[i].record(); ev_side[i].record()
    torch.cuda.synchronize()

    def tail(i, lg, cls, boxes):
        """What only needs the logits (and the boxes): pair geometry, PPN, top-k decode, result gather."""
        # the bbox half of the N^2 pair builder: relative geometry [P, 8, T] of every pair (one lane per
        # (pair, frame), motion channels by wavefront shuffle)
        _, state["geom"] = tspn.ops.pair_gather(None, boxes, pairs, want_feat=False, check_pairs=False)
        _, idx = tspn.ops.ppn_pair_matrix_topk(cls, ppn_w, TOPK_PPN)
        # top-k triplet decode (predict.py:66-106): per pair top-20 of 132, per video top-200
        sc, trip, tid = tspn.ops.decode_topk(lg.view(B, P_vid, K_PRED), local_pairs, cls, row_mul=1,
                                             topk_per_pair=TOPK_PAIR, topk_per_seg=TOPK_SEG, check_pairs=False)
        if use_dist:  # the one collective of the path: final result gather over RCCL
            if args.gather == "decoded":
                state["gathered"] = tspn.dist.gather_decoded(sc, trip, tid, world * B, pair_proposals=idx, force=True)
            else:
                tspn.dist.gather_results(lg.view(B, P_vid, K_PRED), world * B, force=True)
                tspn.dist.gather_results(idx, world * B, force=True)
        state["last"] = (sc, trip, tid, idx)

    def step(i):
        feats, cls, boxes = feats_all[i % nb], cls_all[i % nb], boxes_all[i % nb]
        if bf16:
            if "ws" not in state:   # allocate the workspace once (first warm-up step), then reuse it
                d16 = tspn._abi.FusedBf16Desc()
                d16.B, d16.N, d16.T, d16.D, d16.A, d16.K, d16.P = B, N, T, D, A_ANCH, K_PRED, P
                state["ws"] = torch.empty(tspn._abi.lib().tspn_forward_fused_bf16_workspace_bytes(d16),
                                          dtype=torch.uint8, device=dev)
            _, lg = tspn.ops.forward_fused_bf16(feats, pairs, B, N, packed, conv_b, head_pk, head_b, cls_w,
                                                cls_b, workspace=state["ws"], conv_events=events[i])
            tail(i, lg, cls, boxes)
            return
        lg = out_logits2[i % 2]
        main = torch.cuda.current_stream()
        if overlap and i >= 2:
            main.wait_event(ev_side[i - 2])      # the second stream has finished reading this logits buffer
        # the logits are computed first inside the call (they depend on the tracklet means only) and the
        # event fires there: the second stream decodes and gathers under this step's encoder
        tspn.ops.forward_fused(feats, pairs, B, N, packed, conv_b, head_w, head_b, cls_w, cls_b,
                               workspace=ws, out_heads=out_heads, out_logits=lg, check_pairs=False,
                               conv_events=events[i], canonical_pairs=True,
                               logits_event=ev_logits[i] if overlap else None)
        if not overlap:
            tail(i, lg, cls, boxes)
            return
        with torch.cuda.stream(side):
            side.wait_event(ev_logits[i])
            tail(i, lg, cls, boxes)
            ev_side[i].record(side)

    if via_model:
        # the drop-in surface: BaseModel(cfg) with the same synthetic weights, PairLists in, forward + decode
        cfg = tspn.load_cfg(None, **{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": C,
                                     "PREDICT.FEATURE_DIM": C, "RELPN.DPN.NUM_ANCHORS_PER_LOCATION": A_ANCH,
                                     "PREDICT.PREDICATE_NUM": K_PRED, "RELPN.PPN.NUM_PAIR_PROPOSALS": TOPK_PPN,
                                     "RELPN.DPN.PAIR_GEOMETRY": True,
                                     "RELPN.DPN.CONV_ALGO": "direct" if args.conv == "direct" else "auto"})
        model = tspn.BaseModel(cfg)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model = model.to(dev).eval()

        def step(i):  # noqa: F811
            feats, cls, boxes = feats_all[i % nb], cls_all[i % nb], boxes_all[i % nb]
            plists = [tspn.PairList.from_tracklets(feats[b * N:(b + 1) * N], boxes[b * N:(b + 1) * N], cls[b])
                      for b in range(B)]
            model.profile_conv_events(events[i])
            pair_props, dur_props, rel_logits = model(plists, None)
            dec = model.decode(plists, rel_logits, topk_per_pair=TOPK_PAIR, topk_per_seg=TOPK_SEG)
            sc, trip, tid = (torch.stack([d[k] for d in dec]) for k in range(3))
            idx = torch.stack(pair_props)
            state["geom"] = dur_props[0].geom
            if use_dist:  # the one collective of the path: final result gather over RCCL
                if args.gather == "decoded":
                    state["gathered"] = tspn.dist.gather_decoded(sc, trip, tid, world * B, pair_proposals=idx, force=True)
                else:
                    tspn.dist.gather_results(torch.stack(rel_logits), world * B, force=True)
                    tspn.dist.gather_results(idx, world * B, force=True)
            state["last"] = (sc, trip, tid, idx)

    for i in range(args.warmup):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    props = torch.cuda.get_device_properties(dev)
    clock = ClockSampler("%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                               getattr(props, "pci_device_id", 0)))
    clock.start()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clock_mhz = clock.stop()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if args.gather == "decoded":   # every rank holds every video's decoded rows, in global order
            g = state["gathered"]
            assert g["scores"].shape == (world * B, TOPK_SEG) and g["pair_proposals"].shape == (world * B, TOPK_PPN)
            assert torch.equal(g["scores"][rank * B:(rank + 1) * B], state["last"][0])

    # dominant kernel: the temporal conv of the tracklet projections, HIP events inside the timed steps
    conv_ms = [a.elapsed_time(b) for a, b in events[args.warmup:]]
    conv_avg_s = float(np.mean(conv_ms)) * 1e-3
    conv_flop_direct = 2.0 * (2 * C) * (3 * D) * (B * N * T)  # M=2C, K=3D, columns=B*N*T
    # F(6,3) issues 8 channel-GEMMs on a sixth of the columns (ceil(T/6) sextets per tracklet): 4/9 of the direct work
    frac = {"direct": 1.0, "winograd6": (4.0 / 9.0) * (6 * -(-T // 6)) / T}[args.conv]
    conv_flop = conv_flop_direct * (1.0 if bf16 else frac)
    achieved = conv_flop / conv_avg_s / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS

    if rank == 0:
        pairs_total = world * P * args.steps
        cfg_name = {"cfg2": "BASELINE cfg2: synthetic VidVRD shape N=32 T=150 D=2048 (C=4096, A=4, K=132), fp32, "
                            "random-init weights",
                    "cfg4": "BASELINE cfg4 shard: 64 of the 512 synthetic VidVRD-shaped videos per GPU per step "
                            "(N=32 T=150 D=2048, C=4096, A=4, K=132), fp32, random-init weights",
                    "cfg3": "BASELINE cfg3: VidOR long-clip shape N=64 T=900 D=1024 (C=2048, A=4, K=132), "
                            "bf16 operands / fp32 accumulation, random-init weights"}[args.workload]
        gather_txt = ""
        if use_dist:
            gather_txt = (" + RCCL all-gather of the decoded results (10.8 KB per video)" if args.gather == "decoded"
                          else " + RCCL all-gather of logits/top-k")
        out = {
            "metric": f"tracklet-pairs/sec scored (N={N}, T={T}, D={D})",
            "value": pairs_total / elapsed,
            "unit": "tracklet-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": cfg_name,
                       "videos_per_gpu_per_step": B, "pairs_per_video": P_vid, "resident_input_batches": nb,
                       "surface": ("BaseModel.forward(pair_list) + BaseModel.decode on device-resident PairLists"
                                   if via_model else "ops.forward_fused (pre-allocated workspace and outputs)"),
                       "path": ("fused/factorised (tspn_forward_fused_bf16)" if bf16 else
                                "fused/factorised (tspn_forward_fused_f32)")
                               + " + pair geometry + PPN top-k + top-k triplet decode" + gather_txt
                               + (" (geometry, PPN, decode and gather on a second stream under the encoder)" if overlap
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
        if world == 1 and not args.no_cpu_baseline and not bf16:
            out["cpu_baseline"] = cpu_baseline(wnp, args.cpu_runs)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
