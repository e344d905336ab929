#!/usr/bin/env python3
"""Headline benchmark: tracklet-pairs/sec scored at BASELINE.json cfg2
(synthetic VidVRD shape: N=32 tracklets, T=150 frames, D=2048 RoI dims, fp32).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

`--workload cfg3` runs BASELINE.json's configs[2] instead (N=64, T=900, D=1024, bf16 operands,
4 videos per step) through the same harness; the default (cfg2) is the headline.

A "step" = one pass of the hot path over one batch of `--videos` (default 16) synthetic videos per GPU
(inputs resident in HBM): tracklet tensors -> [pair builder + temporal encoder +
relationness/span heads + RelOIPool + predicate head] (tspn_forward_fused_f32) + PPN
pair-matrix/top-k.  Videos shard across ranks (weak scaling, no collective in the forward);
with N>1 each step ends with ONE RCCL all-gather of the per-pair predicate logits and the
top-k pair indices (the "final result gather").

Printed JSON (rank 0): see the task contract; extras:
  roofline     dominant kernel = the temporal conv of the tracklet projections (fp32 MFMA; Winograd
               F(4,3) by default, --conv winograd2 | direct for the other two algorithms);
               achieved = executed FLOP per launch / HIP-event time of that launch inside the
               timed steps (events recorded on the launch stream by the C ABI's hook).
  cpu_baseline the oracle's reference-faithful dense forward on a bounded sample of pairs,
               all host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

N_TRK, T_FRAMES, D_ROI, A_ANCH, K_PRED = 32, 150, 2048, 4, 132
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (no sparsity), v_mfma_f32_32x32x16_bf16
# --workload cfg3: BASELINE.json configs[2] (VidOR long-clip shape, bf16 operands); not the headline
CFG3 = (64, 900, 1024)
DPN_PRE = "relpn.duration_proposal_network.dpn_head."
PPN_PRE = "relpn.pair_proposal_network.ppn_head."


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos", type=int, default=16, help="videos per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--conv", choices=["winograd4", "winograd2", "winograd", "direct"], default="winograd4",
                    help="temporal-conv algorithm of the tracklet projections (all fp32 MFMA): Winograd F(4,3) "
                         "(default), F(2,3) (= winograd), direct taps")
    ap.add_argument("--workload", choices=["cfg2", "cfg3"], default="cfg2",
                    help="cfg2 = headline (N=32,T=150,D=2048, fp32); cfg3 = N=64,T=900,D=1024 bf16 operands")
    ap.add_argument("--canonical-weights", action="store_true",
                    help="winograd4 only: keep the canonical [6][D][2C] weights (LDS-staged kernel "
                         "conv3_wino43_cl_kernel) instead of the fragment-major default")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the RCCL result gather even with one rank (exercises the N>1 code path)")
    return ap.parse_args()


def cpu_baseline(weights, target_s):
    """Reference-faithful dense forward (oracle.forward_dense: materialise [P,4096,150] ->
    DPNHead -> heads; RelOIPool; predicate head) on a bounded sample of cfg2 pairs."""
    import oracle
    torch.set_num_threads(os.cpu_count() or 1)
    v = tspn.synth.make_video(1, N_TRK, T_FRAMES, D_ROI)
    feats, boxes = torch.from_numpy(v["tracklet_feats"]), torch.from_numpy(v["tracklet_boxes"])
    w = {k: torch.from_numpy(x) for k, x in weights.items()}
    pairs = oracle.pair_index(N_TRK)

    def run(p):
        t0 = time.perf_counter()
        with torch.no_grad():
            oracle.forward_dense(feats, boxes, pairs[:p], w)
        return time.perf_counter() - t0

    run(2)  # warm-up (thread pool, oneDNN primitive cache)
    p_max = N_TRK * (N_TRK - 1)
    p, dt = 8, run(8)
    for _ in range(3):  # stepwise calibration towards ~target_s of CPU work (cost is not linear in p)
        if dt >= 0.6 * target_s or p >= p_max:
            break
        p = int(max(p + 1, min(p_max, p * target_s / max(dt, 1e-3))))
        dt = run(p)
    return {"value": p / dt, "unit": "tracklet-pairs/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{p} of 992 pairs of one cfg2 video, dense reference formulation "
                      f"(oracle.forward_dense), {dt:.1f} s, torch {torch.__version__} CPU"}


def pmc_traffic(videos, conv):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, produced by tools/pmc_summary.py); null if not measured for this
    batch size.  PMC counters cannot be read from inside the process."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        data = json.load(open(path))
        k = data["kernels"][{"winograd4": "conv3_wino43r_kernel", "winograd4c": "conv3_wino43_cl_kernel",
                             "winograd2": "conv3_wino2_cl_kernel"}.get(conv, "conv3_mfma_cl_kernel")]
        if data["videos_per_launch"] == videos:
            return {"traffic": k["hbm_bytes"], "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)",
                    "traffic_source": data["source"]}
    except (OSError, KeyError, ValueError):
        pass
    return {"traffic": None}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: native libraries write there too (RCCL prints a
    # version banner to stdout when the first communicator is created), so fd 1 points at stderr until
    # the result is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    B, N, T, D, C = args.videos, N_TRK, T_FRAMES, D_ROI, 2 * D_ROI
    bf16 = args.workload == "cfg3"
    if bf16:
        N, T, D = CFG3
        C = 2 * D
        B = args.videos if "--videos" in sys.argv else 4
    P_vid = N * (N - 1)

    # ---- weights (seed 0) and inputs (seed 1 + global video index), random-init / synthetic
    sd = tspn.synth.make_weights(0, c=C, a=A_ANCH, k=K_PRED)
    wnp = {"conv_w": sd[DPN_PRE + "conv.weight"], "conv_b": sd[DPN_PRE + "conv.bias"],
           "dur_w": sd[DPN_PRE + "duration_pred.weight"], "dur_b": sd[DPN_PRE + "duration_pred.bias"],
           "rel_w": sd[DPN_PRE + "relness_pred.weight"], "rel_b": sd[DPN_PRE + "relness_pred.bias"],
           "cls_w": sd["classifier.rel_predictor.weight"], "cls_b": sd["classifier.rel_predictor.bias"]}
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    conv_w = d(wnp["conv_w"])
    r16 = lambda x: tspn.ops.cast_bf16(x.contiguous()).float()  # noqa: E731
    if bf16:
        packed = tspn.ops.pack_conv3_bf16(conv_w, split=D)
    else:
        if args.conv == "winograd":
            args.conv = "winograd2"
        packed = {"direct": tspn.ops.pack_conv3, "winograd2": tspn.ops.pack_conv3_wino,
                  "winograd4": tspn.ops.pack_conv3_wino43}[args.conv](conv_w, split=D)
        if args.conv == "winograd4" and not args.canonical_weights:
            packed = tspn.ops.repack_wino43_frag(packed)   # fragment-major: registers-direct kernel
    del conv_w
    conv_b = d(wnp["conv_b"])
    head_w = d(np.concatenate([wnp["rel_w"][:, :, 0], wnp["dur_w"][:, :, 0]]))
    head_b = d(np.concatenate([wnp["rel_b"], wnp["dur_b"]]))
    cls_w, cls_b = d(wnp["cls_w"]), d(wnp["cls_b"])
    if bf16:
        conv_b, head_b, cls_w, cls_b = r16(conv_b), r16(head_b), r16(cls_w), r16(cls_b)
        head_pk = tspn.ops.pack_heads_bf16(head_w)
    ppn_w = {k[len(PPN_PRE):]: d(v) for k, v in sd.items() if k.startswith(PPN_PRE)}

    vids = [tspn.synth.make_video(1 + rank * B + b, N, T, D) for b in range(B)]
    feats = d(np.concatenate([v["tracklet_feats"] for v in vids]))
    if bf16:
        feats = tspn.ops.cast_bf16(feats)
    cls = d(np.stack([v["track_cls_logits"] for v in vids]))
    del vids
    pairs = torch.cat([tspn.ops.pair_index(N, dev, base=b * N) for b in range(B)]).contiguous()
    P = pairs.shape[0]

    ws = None if bf16 else torch.empty(tspn.ops.fused_workspace_bytes(B, N, T, D, A_ANCH, K_PRED, P),
                                       dtype=torch.uint8, device=dev)
    out_heads = torch.empty((P, 3 * A_ANCH, T), dtype=torch.float32, device=dev)
    out_logits = torch.empty((P, K_PRED), dtype=torch.float32, device=dev)
    state = {}
    total_steps = args.warmup + args.steps
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(total_steps)]
    for a, b in events:  # create the HIP event handles
        a.record(); b.record()
    torch.cuda.synchronize()

    def step(i):
        if bf16:
            if "ws" not in state:   # allocate the workspace once (first warm-up step), then reuse it
                d16 = tspn._abi.FusedBf16Desc()
                d16.B, d16.N, d16.T, d16.D, d16.A, d16.K, d16.P = B, N, T, D, A_ANCH, K_PRED, P
                state["ws"] = torch.empty(tspn._abi.lib().tspn_forward_fused_bf16_workspace_bytes(d16),
                                          dtype=torch.uint8, device=dev)
            h16, l16 = tspn.ops.forward_fused_bf16(feats, pairs, B, N, packed, conv_b, head_pk, head_b, cls_w,
                                                   cls_b, workspace=state["ws"], conv_events=events[i])
            state["logits"] = l16
        else:
            tspn.ops.forward_fused(feats, pairs, B, N, packed, conv_b, head_w, head_b, cls_w, cls_b,
                                   workspace=ws, out_heads=out_heads, out_logits=out_logits,
                                   check_pairs=False, conv_events=events[i], canonical_pairs=True)
        _, idx = tspn.ops.ppn_pair_matrix_topk(cls, ppn_w, 256)
        if use_dist:  # the one collective of the path: final result gather over RCCL
            lg = state["logits"] if bf16 else out_logits
            tspn.dist.gather_results(lg.view(B, P_vid, K_PRED), world * B, force=True)
            tspn.dist.gather_results(idx, world * B, force=True)

    for i in range(args.warmup):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # dominant kernel: conv3_mfma (tracklet projections), HIP events inside the timed steps
    conv_ms = [a.elapsed_time(b) for a, b in events[args.warmup:]]
    conv_avg_s = float(np.mean(conv_ms)) * 1e-3
    conv_flop_direct = 2.0 * (2 * C) * (3 * D) * (B * N * T)  # M=2C, K=3D, columns=B*N*T
    # Winograd F(2,3) issues 4 channel-GEMMs on half the columns: 2/3 of the direct MFMA work;
    # F(4,3) issues 6 on a quarter of the columns (ceil(T/4) quads per tracklet): 1/2
    frac = {"direct": 1.0, "winograd2": 2.0 / 3.0, "winograd4": 0.5 * (4 * -(-T // 4)) / T}[args.conv]
    conv_flop = conv_flop_direct * (1.0 if bf16 else frac)
    achieved = conv_flop / conv_avg_s / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS

    if rank == 0:
        pairs_total = world * P * args.steps
        out = {
            "metric": f"tracklet-pairs/sec scored (N={N}, T={T}, D={D})",
            "value": pairs_total / elapsed,
            "unit": "tracklet-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE cfg3: VidOR long-clip shape N=64 T=900 D=1024 (C=2048, A=4, K=132), "
                                    "bf16 operands / fp32 accumulation, random-init weights" if bf16 else
                                    "BASELINE cfg2: synthetic VidVRD shape N=32 T=150 D=2048 (C=4096, A=4, "
                                    "K=132), fp32, random-init weights"),
                       "videos_per_gpu_per_step": B, "pairs_per_video": P_vid,
                       "path": ("fused/factorised (tspn_forward_fused_bf16) + PPN top-k" if bf16 else
                                "fused/factorised (tspn_forward_fused_f32) + PPN top-k")
                               + (" + RCCL all-gather of logits/top-k" if use_dist else ""),
                       "dense_equivalent_gflop_per_pair": (2.0 * T * C * (3 * C + 3 * A_ANCH) + 2.0 * C * K_PRED) / 1e9,
                       "conv_algo": "direct" if bf16 else args.conv,
                       "executed_gflop_per_pair": (conv_flop + 2.0 * P * T * C * 16 + 2.0 * P * C * K_PRED) / P / 1e9},
            "roofline": {"bound": "mfma",
                         "kernel": ("conv3_bf16_big_kernel (tracklet projections: k=3 conv as bf16 32x32x16 MFMA "
                                    "implicit GEMM, M=2C, K=3D)" if bf16 else
                                    ("conv3_wino43_cl_kernel" if args.canonical_weights else "conv3_wino43r_kernel") +
                                    " (tracklet projections: k=3 conv, Winograd F(4,3), "
                                    "fp32 32x32x2 MFMA, M=2C, 6 channel-GEMMs of K=D on a quarter of the columns)"
                                    if args.conv == "winograd4" else
                                    "conv3_wino2_cl_kernel (tracklet projections: k=3 conv, Winograd F(2,3), "
                                    "fp32 32x32x2 MFMA, M=2C, 4 channel-GEMMs of K=D on half the columns)"
                                    if args.conv == "winograd2" else
                                    "conv3_mfma_cl_kernel (tracklet projections: k=3 conv as fp32 32x32x2 "
                                    "MFMA implicit GEMM, M=2C, K=3D)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, **({"traffic": None} if bf16 else pmc_traffic(
                             B, "winograd4c" if (args.conv == "winograd4" and args.canonical_weights) else args.conv)),
                         "direct_equivalent_tflops": conv_flop_direct / conv_avg_s / 1e12,
                         "flop_per_launch": conv_flop, "avg_launch_ms": conv_avg_s * 1e3,
                         "share_of_step": conv_avg_s / (elapsed / args.steps)},
        }
        if world == 1 and not args.no_cpu_baseline and not bf16:
            out["cpu_baseline"] = cpu_baseline(wnp, args.cpu_seconds)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
