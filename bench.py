#!/usr/bin/env python
"""Benchmark of the MFCNet hot path on MI355X: frames/s of the full training step
(zero_grad -> forward -> fused log_softmax/NLL/soft-Jaccard -> backward -> Adam; src/engine.py:54-71)
on synthetic 480x640 T=3 clips, BASELINE.json's metric.

    python bench.py                                   # 1 GPU, width 32 (the metric's "HRNet-w32"), bf16, B=8
    python bench.py --width 48                        # the reference's hard-coded HRNet-W48
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # one rank per GPU, RCCL all-reduce of the flat gradient arena

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the
dominant kernel (the MFMA implicit-GEMM convolution) inside the timed steps; `cpu_baseline` times the CPU
oracle (the restatement of the reference step, oracle/mfcnet_oracle.py) on this host's cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBPS = 8000.0                              # HBM3E, MI355X_MICROARCH.md
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
NT_SLOTS = [1, 2, 3, 4, 6]


def synth(B, T, H, W, nc, seed, device, depth=False, optflow=False):
    """N(0,1) frames, U[0,1) depth maps, 3*N(0,1) pixel-unit flows (constants), uniform class masks."""
    g = torch.Generator().manual_seed(seed)
    frames = [torch.randn(B, 3, H, W, generator=g).to(device) for _ in range(T)]
    mask = torch.randint(0, nc, (B, H, W), generator=g).to(device)
    dm = [torch.rand(B, 1, H, W, generator=g).to(device) for _ in range(T)] if depth else None
    fl = [(3.0 * torch.randn(B, 2, H, W, generator=g)).to(device) for _ in range(T - 1)] if optflow else None
    return frames, mask, dm, fl


def pmc_traffic(NT, MT, PM, NW, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile_round.sh: two separate
    rocprofv3 --pmc runs of this command, FETCH_SIZE doubled per MI355X_MICROARCH.md; condensed by
    tools/summarize_profile.py).  bench.py cannot collect counters itself; null when no profile matches the configuration."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_w{args.width}_pmc_traffic.json")))      # newest round's passes
    if args.dtype != "bf16" or (args.batch, args.frames, args.height, args.width_px) != (8, 3, 480, 640) or not found \
            or args.depth or args.optflow or args.basic or args.single or args.fwd_only:
        return None, None
    path = found[-1]
    tag = os.path.basename(path)
    with open(path) as f:
        ks = json.load(f)["kernels"]
    for name, v in ks.items():
        if "conv_igemm_kernel" in name and f"Li{NT}ELi{MT}ELi{PM}ELi{NW}E" in name and "DF16b" in name:
            return v["hbm_bytes_per_launch"], f"profiles/{tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --serial`)"
    return None, None


def cpu_baseline(width, T, H, W, steps=2):
    """The CPU oracle's training step on the host cores (bounded sample)."""
    from oracle import mfcnet_oracle as O
    torch.manual_seed(0)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))     # B=1 convolutions stop scaling (and oversubscribe) beyond ~32 threads
    sd = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", width, 5, T, False, False))
    net = O.Net(sd, "HRNetMulti-Large", width, 5, T).train()
    opt = O.make_adam(net, 1e-4)
    g = torch.Generator().manual_seed(42)
    frames = [torch.randn(1, 3, H, W, generator=g) for _ in range(T)]
    mask = torch.randint(0, 5, (1, H, W), generator=g)
    O.train_step(net, opt, frames, mask)            # warm-up (allocator, thread pool)
    t0 = time.time()
    for _ in range(steps):
        O.train_step(net, opt, frames, mask)
    dt = time.time() - t0
    return {"value": round(steps * T / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full training steps of the CPU oracle, HRNet-w{width} MFCNet, B=1, T={T}, {H}x{W}, fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=32, help="HRNet width: 32 = BASELINE.json metric label, 48 = reference")
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width-px", type=int, default=640)
    ap.add_argument("--depth", action="store_true", help="add the T depth maps (BASELINE.json configs[3])")
    ap.add_argument("--optflow", action="store_true", help="add the T-1 optical-flow fields (BASELINE.json configs[3])")
    ap.add_argument("--basic", action="store_true", help="model_type HRNetMulti-Basic (flow warp) instead of HRNetMulti-Large")
    ap.add_argument("--single", action="store_true", help="single-frame HRNet (model_type 'HRNet'): one [B,3,H,W] tensor in (BASELINE.json configs[1])")
    ap.add_argument("--fwd-only", action="store_true", help="eval-mode forward only (no loss / backward / optimizer)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--serial", action="store_true", help="one stream: no branch lanes / detached weight-gradient streams (for kernel profiles)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()          # (gloo rehearsal: several ranks may share one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            # RCCL's stream from the high-priority pool: those streams have hardware queues of their own, so the bucket
            # all-reduces can never land on the hardware queue of the backward chain ("which streams?")
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
            except Exception:
                opts = None
            dist.init_process_group("nccl", device_id=device, pg_options=opts)
        else:
            dist.init_process_group(args.backend)

    if os.environ.get("MFC_DUMMY_STREAMS"):         # experiment: streams created (and used) by others before ours shift the HW-queue mapping
        _dummies = [torch.cuda.Stream() for _ in range(int(os.environ["MFC_DUMMY_STREAMS"]))]
        for ds in _dummies:
            with torch.cuda.stream(ds):
                torch.zeros(16, device=device).add_(1)
        torch.cuda.synchronize()
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.dist import GradBucketReducer

    if args.single:
        args.frames = 1
    T, H, W, B, nc = args.frames, args.height, args.width_px, args.batch, 5
    torch.manual_seed(1234)                                    # identical initial weights on every rank
    cls = mfc.HRNetMultiBasic if args.basic else mfc.HRNetMultiLarge
    if args.single:
        model = mfc.HighResolutionNetHIP(num_classes=nc, width=args.width, compute_dtype=args.dtype)
    else:
        model = cls(num_classes=nc, num_frames=T, pretrained=False, width=args.width, compute_dtype=args.dtype,
                    optflow_inputs=args.optflow, depth_inputs=args.depth)
    model = model.to(device)
    model = model.eval() if args.fwd_only else model.train()
    if os.environ.get("MFC_MASK_BITS"):            # tuning: 0 = the BatchNorm backward reads the ReLU mask from the bf16 tensor
        model.relu_mask_bits = os.environ["MFC_MASK_BITS"] != "0"
    if os.environ.get("MFC_BATCH_WGRAD"):          # tuning: 0 = one launch per weight gradient
        model.batch_wgrad = os.environ["MFC_BATCH_WGRAD"] != "0"
    opt = mfc.FlatAdam(model, lr=1e-4)
    frames, mask, depth, flow = synth(B, T, H, W, nc, 42 + 2000 + rank, device, args.depth, args.optflow)

    # (MFC_FORCE_BUCKETS: run the segmented backward + bucket hook on one GPU too, to measure what the segmentation costs)
    reducer = GradBucketReducer(model, average=False) if (world > 1 or os.environ.get("MFC_FORCE_BUCKETS")) else None

    def fwd():
        return model(frames[0]) if args.single else model(frames, optflow=flow, depth=depth)

    def step():
        if args.fwd_only:
            with torch.no_grad():
                return fwd().float().mean()
        opt.zero_grad()
        out = fwd()
        loss, _ = mfc.mfc_loss(out, mask, global_batch=True)      # loss over the global batch (all-reduce of 26 sums), as the reference
        loss.backward()                                             # (per-bucket all-reduces start inside, next to the backward kernels)
        if reducer is not None:
            reducer.finish()                                        # ... so the ranks' gradients add up (SUM: global-batch loss)
        opt.step()
        return loss

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serial:
        L.lib.mfc_set_flag(9, 0)
    if os.environ.get("MFC_WGRAD_BLOCKS"):          # (must be set before the plan is built: it sizes the partial-sum slices)
        L.lib.mfc_set_flag(11, int(os.environ["MFC_WGRAD_BLOCKS"]))
    if os.environ.get("MFC_BNRED_BLOCKS"):
        L.lib.mfc_set_flag(27, int(os.environ["MFC_BNRED_BLOCKS"]))
    if os.environ.get("MFC_CONV_GEMM"):
        L.lib.mfc_set_flag(23, int(os.environ["MFC_CONV_GEMM"]))
    if os.environ.get("MFC_PROBE_STREAMS"):
        L.lib.mfc_set_flag(22, int(os.environ["MFC_PROBE_STREAMS"]))
    if os.environ.get("MFC_WGRAD_MAXPX"):           # (before the plan is built, like MFC_WGRAD_BLOCKS)
        L.lib.mfc_set_flag(21, int(os.environ["MFC_WGRAD_MAXPX"]))
    if os.environ.get("MFC_CONV_NW8"):              # weight (%) of the 8-wave conv geometries in the search (0 = never); before the plan is built
        L.lib.mfc_set_flag(19, int(os.environ["MFC_CONV_NW8"]))
    if os.environ.get("MFC_CONV_FILL_PCT"):         # before the plan is built (the packed weight layouts depend on the geometry)
        L.lib.mfc_set_flag(18, int(os.environ["MFC_CONV_FILL_PCT"]))
    if os.environ.get("MFC_CONV_GRID"):             # before the plan is built (sizes nothing, but the layouts are queried then)
        L.lib.mfc_set_flag(4, int(os.environ["MFC_CONV_GRID"]))
    if os.environ.get("MFC_ASYNC_PRIO"):            # before the first program run (read when the streams are created)
        L.lib.mfc_set_flag(16, int(os.environ["MFC_ASYNC_PRIO"]))
    if os.environ.get("MFC_SKIP_KINDS"):            # what-if timing only (results are wrong)
        L.lib.mfc_set_flag(15, int(os.environ["MFC_SKIP_KINDS"]))
    if os.environ.get("MFC_ASYNC_ON_LANE"):
        L.lib.mfc_set_flag(13, int(os.environ["MFC_ASYNC_ON_LANE"]))
    if os.environ.get("MFC_OWN_MAIN"):
        L.lib.mfc_set_flag(14, int(os.environ["MFC_OWN_MAIN"]))
    if os.environ.get("MFC_LANE_STREAMS"):
        L.lib.mfc_set_flag(12, int(os.environ["MFC_LANE_STREAMS"]))
    if os.environ.get("MFC_LANES"):
        L.lib.mfc_set_flag(9, int(os.environ["MFC_LANES"]))
    if os.environ.get("MFC_ASYNC_STREAMS"):
        L.lib.mfc_set_flag(10, int(os.environ["MFC_ASYNC_STREAMS"]))
    for _ in range(args.warmup):
        step()
    sync()
    # HIP events bracket every conv / wgrad launch of the LAST timed step only: recording events around ~1100 launches
    # per step costs ~7 % of a step (it breaks back-to-back dispatch), so one profiled step keeps `value` honest while the
    # roofline is still measured live inside the timed region, on the launch stream.
    prof_steps = 0 if args.no_prof else 1
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - prof_steps:
            # the profiled step runs every record on ONE stream: with the branch lanes / detached wgrad streams a launch shares
            # the GPU with other kernels and its event-bracketed time would no longer be the kernel's own duration
            L.lib.mfc_set_flag(9, 0)
            L.lib.mfc_prof_enable(1)
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    L.lib.mfc_prof_enable(0)
    L.lib.mfc_set_flag(9, 0 if args.serial else 3)
    prof = L.ProfResult()
    L.lib.mfc_prof_collect(C.byref(prof))
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss.detach())

    extra = ("+depth" if args.depth else "") + ("+optflow" if args.optflow else "")
    cfg_label = ("per-GPU share of BASELINE.json configs[3]" if (T, H, W, B, extra) == (3, 480, 640, 4, "+depth+optflow") else
                 "BASELINE.json configs[1]" if (args.single, args.fwd_only, H, W, B) == (True, True, 480, 640, 8) else
                 "custom inputs" if (extra or args.single or args.fwd_only) else
                 "BASELINE.json configs[2]" if (T, H, W, B) == (3, 480, 640, 8) else
                 "per-GPU share of BASELINE.json configs[4]" if (T, H, W, B, args.width) == (5, 720, 960, 8, 48) else "custom size")
    if rank == 0:
        # dominant kernel = the conv_igemm instantiation (one profiler bucket per <NT, MT, PMAX>, as rocprof names them) with the
        # largest total time in the profiled step; bucket layout: include/mfcnet_hip.h (mfc_prof_result)
        dt_base = 64 if args.dtype == "bf16" else 0
        roof = None
        conv = [(dt_base + s, s) for s in range(41)]          # 40 = conv_gemm1x1_kernel (big 1x1 convolutions)
        live = [(b, s) for b, s in conv if prof.launches[b]]
        if live:
            best, bslot = max(live, key=lambda bs: prof.ms[bs[0]])
            n = prof.launches[best]
            avg_ms = prof.ms[best] / n
            achieved = prof.flops[best] / n / (avg_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[args.dtype]
            if bslot == 40:
                NT, MT, PM, NW = 4, 8, 0, 8
                kname = "conv_gemm1x1_kernel"
            else:
                NW = 8 if bslot >= 20 else 4
                NT = NT_SLOTS[(bslot % 20) // 4]
                MT, PM = [(4, 3), (4, 6), (2, 4), (2, 10)][bslot % 4]
                kname = f"conv_igemm_kernel<{'__bf16' if args.dtype == 'bf16' else 'float'}, {NT}, {MT}, {PM}, {NW}>"
            fam_ms = sum(prof.ms[b] for b, _ in conv)
            fam_fl = sum(prof.flops[b] for b, _ in conv)
            wgb = [128 + dt_base + s for s in range(64)]
            wg_ms = sum(prof.ms[b] for b in wgb)
            wg_fl = sum(prof.flops[b] for b in wgb)
            traffic, tsrc = pmc_traffic(NT, MT, PM, NW, args)
            # which roof binds this kernel: its algorithmic bytes at the HBM peak vs its flops at the dense MFMA peak
            abytes = prof.bytes[best] / n
            gbps = abytes / (avg_ms * 1e-3) / 1e9
            hbm_bound = abytes / (PEAK_HBM_GBPS * 1e9) > (prof.flops[best] / n) / (peak * 1e12)
            roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
                    "achieved": round(gbps if hbm_bound else achieved, 2), "peak": PEAK_HBM_GBPS if hbm_bound else peak,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(gbps / PEAK_HBM_GBPS if hbm_bound else achieved / peak, 4),
                    "mfma_tflops": round(achieved, 2), "mfma_frac": round(achieved / peak, 4),
                    "hbm_gbps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4),
                    "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": round(prof.bytes[best] / n),
                    "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": n // max(prof_steps, 1),
                    "flops_per_launch": prof.flops[best] / n,
                    "all_conv_igemm_tflops": round(fam_fl / (fam_ms * 1e-3) / 1e12, 2) if fam_ms else None,
                    "all_wgrad_tflops": round(wg_fl / (wg_ms * 1e-3) / 1e12, 2) if wg_ms else None,
                    "conv_ms_per_step": round(fam_ms / max(prof_steps, 1), 3), "wgrad_ms_per_step": round(wg_ms / max(prof_steps, 1), 3),
                    "profiled_steps": prof_steps, "profiled_step_streams": "serial"}
        out = {"metric": f"frames/sec ({H}x{W}, T={T}, HRNet {'single-frame' if args.single else 'MFCNet'}) {'fwd' if args.fwd_only else 'fwd+bwd'}", "value": round(world * B * T * args.steps / dt, 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{'single-frame HRNet' if args.single else 'MFCNet'}{'-Basic' if args.basic else ''} T={T} {'RGB' + extra if extra else 'RGB-only'} (HRNet-w{args.width} base), {H}x{W}, batch={B}/GPU, {'eval forward only' if args.fwd_only else 'fwd+bwd+Adam'} "
                                      f"({cfg_label})", "width": args.width, "global_batch": world * B,
                          "frames_per_clip": T, "parallelism": f"dp{world}", "streams": "serial" if args.serial else "branch lanes + detached wgrad", "final_loss": round(final_loss, 5)},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.width, T, H, W)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
