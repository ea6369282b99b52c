#!/usr/bin/env python
"""Benchmark of the MFCNet hot path on MI355X: frames/s of the full training step
(zero_grad -> forward -> fused log_softmax/NLL/soft-Jaccard -> backward -> Adam; src/engine.py:54-71)
on synthetic 480x640 T=3 clips, BASELINE.json's metric.

    python bench.py                                   # 1 GPU, width 32 (the metric's "HRNet-w32"), bf16, B=8
    python bench.py --width 48                        # the reference's hard-coded HRNet-W48
    python bench.py --gpus N                          # starts N ranks itself (one per GPU, RCCL), relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # the same under an external launcher

Prints ONE JSON line (rank 0).  `value` / `ms_per_step` come from the wall clock around EXACTLY the K timed steps (barrier +
synchronize on both sides, MAX over ranks); every timed step is also bracketed by a HIP event pair on the step's stream and
`step_ms` gives the median / min / max of those K durations (`value_median` = frames/s at the median step).
`roofline` describes the DOMINANT KERNEL TEMPLATE (all instantiations of one `__global__` template summed: the convolution
kernel has a dozen) of a one-stream profiled step that runs AFTER the timed region (HIP events around every launch, on the launch
stream; rows named as rocprofv3 names them), and carries the whole-step figures: `step_frac` = sum over all launches of the time
their algorithmic bytes / FLOPs take at the HBM / MFMA peak, divided by the measured step; `step.mfma_frac`, `step.hbm_frac`
(algorithmic and counter bytes), and the counter bytes against the fused minimum of SURVEY.md 8(d).  `roofline_top` is the
single instantiation with the largest total time, `roofline_conv` the dominant convolution instantiation; `cpu_baseline` times
the CPU oracle (the restatement of the reference step, oracle/mfcnet_oracle.py) on this host's cores (median of >= 5 steps).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0                              # HBM3E, MI355X_MICROARCH.md
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def parse_args():
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="storage type of activations / packed weights (fp16: IEEE half with a static loss scale, BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the profiled step (no `roofline` object)")
    ap.add_argument("--fp16-line", action="store_true", help="also measure the same step with fp16 storage (`fp16_storage` object; opt-in since round 4: "
                    "it builds and times a second model)")
    ap.add_argument("--no-fp16-line", action="store_true", help="(accepted for old scripts: the fp16 side line is off unless --fp16-line)")
    ap.add_argument("--no-fp32-parity", action="store_true", help="skip the `fp32_parity` side object (throughput of the fp32 parity mode + max |logits - oracle| "
                    "on one clip; needs the CPU baseline leg, which evaluates the oracle's side of it)")
    ap.add_argument("--sharded-adam", action="store_true", help="N > 1: reduce-scatter of the gradient arena + Adam on the rank's shard + all-gather of the "
                    "parameters (dist.ShardedStep) instead of the bucketed all-reduce overlapped with the backward + replicated Adam")
    ap.add_argument("--weights", default="hash", choices=["hash", "init"], help="hash = key-hash generator of SURVEY.md 8(c)/(d) (the weights of the golden "
                    "fixtures; default), init = torch's default initialisation from a fixed seed")
    ap.add_argument("--serial", action="store_true", help="one stream: no branch lanes / detached weight-gradient streams (for kernel profiles)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the data-parallel step path even with one rank "
                    "(exercises the RCCL code path -- process-group options, bucket all-reduces, loss-sum all-reduce -- on a one-GPU box)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- N ranks from one command
def launch_ranks(args):
    """`bench.py --gpus N` without an external launcher: this parent never touches the GPU (no HIP call, torch is not even
    imported); it starts N fresh ranks under torch.distributed.run on 127.0.0.1, relays rank 0's JSON line and fails unless
    that line reports n_gpus == N (the replacement of the reference's single-process nn.DataParallel,
    scripts/train_multiframe_detection.py:107-110)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MFC_BENCH_LAUNCHER="bench.py")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (rc {r.returncode}, {len(lines)} result lines)")
    d = json.loads(lines[0])
    if d.get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} GPUs, the ranks report {d.get('n_gpus')}")
    print(lines[0], flush=True)


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def synth(B, T, H, W, nc, seed, device, depth=False, optflow=False):
    """N(0,1) frames, U[0,1) depth maps, 3*N(0,1) pixel-unit flows (constants), uniform class masks."""
    import torch
    g = torch.Generator().manual_seed(seed)
    frames = [torch.randn(B, 3, H, W, generator=g).to(device) for _ in range(T)]
    mask = torch.randint(0, nc, (B, H, W), generator=g).to(device)
    dm = [torch.rand(B, 1, H, W, generator=g).to(device) for _ in range(T)] if depth else None
    fl = [(3.0 * torch.randn(B, 2, H, W, generator=g)).to(device) for _ in range(T - 1)] if optflow else None
    return frames, mask, dm, fl


def code_hash():
    """sha1 over what decides the launches and their traffic (kernel sources, headers, the planner): a committed PMC table is only as good as the
    code it was taken on -- tools/summarize_profile.py stores this hash in the table, bench.py says whether it still matches."""
    import glob, hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "mfcnet-tracker_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mfcnet-tracker_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "mfcnet-tracker_amd", "mfcnet_amd", "plan.py")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/profile_round.sh: separate rocprofv3 --pmc runs of
    this command, FETCH_SIZE doubled per MI355X_MICROARCH.md; condensed by tools/summarize_profile.py).  bench.py cannot
    collect counters itself; null when no committed profile matches the configuration."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_w{args.width}_pmc_traffic.json")))      # newest round's passes
    if args.dtype != "bf16" or (args.batch, args.frames, args.height, args.width_px) != (8, 3, 480, 640) or not found \
            or args.depth or args.optflow or args.basic or args.single or args.fwd_only:
        return None, None
    path = found[-1]
    with open(path) as f:
        doc = json.load(f)
    fresh = doc.get("code_hash") == code_hash()
    src = f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --serial`; " + \
          ("taken on this code: kernel sources, headers and planner hash " + str(doc.get("code_hash")) if fresh else
           "STALE: taken on other kernel / planner sources than the ones running (hash " + str(doc.get("code_hash")) + " vs " + code_hash() + ")") + ")"
    if kernel is None:                                        # whole step: counter bytes of every kernel of one training step
        return (doc.get("step") or {}).get("hbm_bytes_per_step"), src
    ks = doc["kernels"]
    if isinstance(kernel, (list, tuple)):                     # a template family: launch-weighted mean over its instantiations
        hit = [(ks[k]["hbm_bytes_per_launch"], n) for k, n in kernel if k in ks]
        if len(hit) != len(kernel) or not hit:
            return None, None
        return round(sum(b * n for b, n in hit) / sum(n for _, n in hit)), src
    v = ks.get(kernel)
    if v is None:
        return None, None
    return v["hbm_bytes_per_launch"], src


# ----------------------------------------------------------------------------------------------- CPU baseline (rank 0, N = 1)
def host_cpu():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(width, T, H, W, parity_clip=False, optflow=False, depth=False):
    """The CPU oracle on the host cores, bounded samples: the MFCNet training step (the value), its
    eval-mode forward, and the single-frame HRNet B=2 training step that stands in for BASELINE configs[0]."""
    import torch
    from oracle import mfcnet_oracle as O
    torch.manual_seed(0)
    model, logical, usable = host_cpu()
    threads = min(usable, 32)                     # B=1 convolutions stop scaling beyond ~32 threads (measured on the 64-core EPYC 9575F host: 1.1 frames/s at 32, 0.43 at 64)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(42)

    def timed(fn, n):
        """median of n timed calls after one warm-up call (allocator, thread pool)"""
        fn()
        ts = []
        for _ in range(n):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        ts.sort()
        return ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])

    def mfc_step_time(Bc, n):
        sd = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", width, 5, T, False, False))
        net = O.Net(sd, "HRNetMulti-Large", width, 5, T).train()
        opt = O.make_adam(net, 1e-4)
        frames = [torch.randn(Bc, 3, H, W, generator=g) for _ in range(T)]
        mask = torch.randint(0, 5, (Bc, H, W), generator=g)
        t_step = timed(lambda: O.train_step(net, opt, frames, mask), n)
        t_fwd = None
        if Bc == 1:
            net.eval()
            with torch.no_grad():
                t_fwd = timed(lambda: net(frames), n)
        return t_step, t_fwd

    t_step, t_fwd = mfc_step_time(1, 5)
    t_step2, _ = mfc_step_time(2, 3)
    # BASELINE configs[0]: single-frame model, batch 2, full step (`F.log_softmax(model(x))`, scripts/train_toolpose_segmentation.py:162-163);
    # its "UNet" model type does not exist in the reference -- the single-frame HRNet-W48 stands in
    sds = O.hashed_state(O.hrnet_table(48, 5, ""))
    sn = O.SingleNet(sds, 48, 5).train()
    so = torch.optim.Adam(sn.params(""), lr=1e-4)
    xs = torch.randn(2, 3, H, W, generator=g)
    ms = torch.randint(0, 5, (2, H, W), generator=g)

    def single_step():
        so.zero_grad(set_to_none=True)
        loss, _ = O.total_loss(sn(xs), ms, 5)
        loss.backward()
        so.step()
    t_single = timed(single_step, 3)
    parity = None
    if parity_clip:
        # the oracle's half of `fp32_parity`: eval-mode logits of one hash-generated clip with the key-hash weights (the GPU side runs the same clip)
        sdp = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", width, 5, T, optflow, depth))
        netp = O.Net(sdp, "HRNetMulti-Large", width, 5, T, optflow, depth).eval()
        pf, pfl, pd, _ = O.synthetic_clip("bench/parity", 1, T, H, W, optflow, depth)
        with torch.no_grad():
            parity = {"inputs": (pf, pfl, pd), "logits": netp(pf, pfl, pd).float()}
    return {"value": round(T / t_step, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "host": {"cpu_model": model, "logical_cpus": logical, "usable_cpus": usable, "torch_threads": threads,
                     "threads_note": "32 of the host's cores: the B=1 convolutions of torch's CPU backend are SLOWER with more (measured 1.1 frames/s at 32 threads, 0.43 at 64)"},
            "sample": f"median of 5 full training steps (after 1 warm-up) of the CPU oracle (fwd + loss + bwd + Adam), HRNet-w{width} MFCNet, B=1, T={T}, {H}x{W}, fp32",
            "batch2": {"value": round(2 * T / t_step2, 4), "unit": "frames/s", "sample": f"median of 3 training steps of the same model at B=2 (the GPU side runs B=8 per GPU; "
                       "the oracle keeps every activation of the autograd graph: ~8 GB per clip at fp32)"},
            "fwd_only": {"value": round(T / t_fwd, 4), "unit": "frames/s", "sample": f"median of 5 eval-mode forwards of the same clip (B=1, T={T})"},
            "single_frame_b2_step": {"value": round(2 / t_single, 4), "unit": "frames/s",
                                     "sample": f"median of 3 full training steps of the single-frame HRNet-w48 (model_type 'HRNet'), B=2, {H}x{W}, fp32 "
                                               "(stand-in for BASELINE.json configs[0])"}}, parity


def roofline_of(row, steps, dtype, traffic=(None, None)):
    """`roofline` object of one profiler row: which roof binds it (its algorithmic bytes at the HBM peak vs its flops at the
    dense MFMA peak -- whichever takes longer) and the fraction of that roof it reaches."""
    n = row["launches"]
    avg_ms = row["ms"] / n
    tfl = row["flops"] / n / (avg_ms * 1e-3) / 1e12
    gbps = row["bytes"] / n / (avg_ms * 1e-3) / 1e9
    peak = PEAK_TFLOPS[dtype]
    hbm_bound = row["bytes"] / (PEAK_HBM_GBPS * 1e9) >= row["flops"] / (peak * 1e12)
    return {"bound": "hbm" if hbm_bound else "mfma", "kernel": row["name"],
            "achieved": round(gbps if hbm_bound else tfl, 2), "peak": PEAK_HBM_GBPS if hbm_bound else peak,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(gbps / PEAK_HBM_GBPS if hbm_bound else tfl / peak, 4),
            "mfma_tflops": round(tfl, 2), "mfma_frac": round(tfl / peak, 4), "hbm_gbps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4),
            "traffic": traffic[0], "traffic_source": traffic[1], "algorithmic_bytes_per_launch": round(row["bytes"] / n),
            "flops_per_launch": row["flops"] / n, "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": n // max(steps, 1),
            "ms_per_step": round(row["ms"] / max(steps, 1), 3)}


def main():
    args = parse_args()
    external = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not external:
        return launch_ranks(args)                 # (before anything touches the GPU)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); they must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    if local >= ndev and args.backend == "nccl":
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own (LOCAL_RANK {local}, {ndev} visible); one rank per GPU")
    local = local % ndev                               # (gloo rehearsal: several ranks may share one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    devs = [(socket.gethostname(), local)]
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_PORT" not in os.environ:                 # (--force-dist without a launcher: a free port, not a fixed one)
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        if args.force_dist:
            os.environ["MFC_DIST_FORCE"] = "1"
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
            try:
                dist.init_process_group("nccl", device_id=device, pg_options=opts)
            except (TypeError, ValueError):          # (an older / newer torch that does not take these keywords: plain initialisation)
                if dist.is_initialized():
                    raise
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")
        devs = [None] * world
        dist.all_gather_object(devs, (socket.gethostname(), local))
        if args.backend == "nccl" and len(set(devs)) != world:
            raise SystemExit(f"bench.py: ranks share a device: {devs}")

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
    if args.weights == "hash":
        from mfcnet_amd.synth import fill_hashed
        fill_hashed(model)                                      # every state_dict entry a pure function of its key
    model = model.to(device)
    model = model.eval() if args.fwd_only else model.train()
    for env, attr in (("MFC_MASK_BITS", "relu_mask_bits"), ("MFC_BATCH_WGRAD", "batch_wgrad"), ("MFC_FUSE_BNRED", "fuse_bnbwd_reduce"), ("MFC_MERGE_S2", "merge_s2_dgrad"), ("MFC_ONE_JOIN", "fuse_one_join")):      # tuning switches of the plan
        if os.environ.get(env):
            setattr(model, attr, os.environ[env] != "0")
    opt = mfc.FlatAdam(model, lr=1e-4)
    frames, mask, depth, flow = synth(B, T, H, W, nc, 42 + 2000 + rank, device, args.depth, args.optflow)

    # (MFC_FORCE_BUCKETS: run the segmented backward + bucket hook on one GPU too, to measure what the segmentation costs)
    reducer = GradBucketReducer(model, average=False) if ((world > 1 or args.force_dist or os.environ.get("MFC_FORCE_BUCKETS")) and not args.sharded_adam) else None
    sharded = None
    if args.sharded_adam and (world > 1 or args.force_dist):
        from mfcnet_amd.dist import ShardedStep
        sharded = ShardedStep(model)

    def fwd():
        return model(frames[0]) if args.single else model(frames, optflow=flow, depth=depth)

    lscale = 1.0 if args.dtype != "fp16" else mfc.engine.loss_scale_for(model, torch.empty(B * max(world, 1), nc, H, W, device="meta"))

    def step():
        if args.fwd_only:
            with torch.no_grad():
                return fwd().float().mean()
        opt.zero_grad()
        out = fwd()
        loss, _ = mfc.mfc_loss(out, mask, global_batch=True)      # loss over the global batch (all-reduce of 26 sums), as the reference
        (loss * lscale if lscale != 1.0 else loss).backward()       # (per-bucket all-reduces start inside, next to the backward kernels)
        if reducer is not None:
            reducer.finish()                                        # ... so the ranks' gradients add up (SUM: global-batch loss)
        if sharded is not None:
            sharded.step(opt, grad_scale=1.0 / lscale)              # reduce-scatter -> Adam on this rank's shard -> all-gather
        else:
            opt.step(grad_scale=1.0 / lscale)
        return loss

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # tuning switches of the library (defaults are the measured optima; include/mfcnet_hip.h lists them).  Set before the plan is built.
    for env, flag in (("MFC_WGRAD_BLOCKS", 11), ("MFC_BNRED_BLOCKS", 27), ("MFC_CONV_GEMM", 23), ("MFC_PROBE_STREAMS", 22),
                      ("MFC_WGRAD_MAXPX", 21), ("MFC_CONV_NW8", 19), ("MFC_CONV_FILL_PCT", 18), ("MFC_CONV_GRID", 4),
                      ("MFC_ASYNC_PRIO", 16), ("MFC_SKIP_KINDS", 15), ("MFC_ASYNC_ON_LANE", 13), ("MFC_OWN_MAIN", 14),
                      ("MFC_LANE_STREAMS", 12), ("MFC_LANES", 9), ("MFC_ASYNC_STREAMS", 10), ("MFC_WGRAD_DMA", 29), ("MFC_CONV_MT", 2), ("MFC_CONV_LDS_KB", 6),
                      ("MFC_CONV_RING", 30), ("MFC_RING_WGS", 33), ("MFC_WGRAD_XF8", 38), ("MFC_APPLYFIN_BLOCKS", 39), ("MFC_BNRED_THREADS", 41), ("MFC_BNRED_MINPX", 42), ("MFC_WGRAD_DMA_S2", 46), ("MFC_WGRAD_DMA48", 47), ("MFC_WGRAD_DMA48_X2", 48), ("MFC_WT_MIN_MB", 54), ("MFC_CONV_NW8_FUSED", 55), ("MFC_LANE4_FWD", 56), ("MFC_RING_C64_UNFUSED_KPX", 57)):
        if os.environ.get(env):
            L.lib.mfc_set_flag(flag, int(os.environ[env]))
    if args.serial:
        L.lib.mfc_set_flag(9, 0)
    for _ in range(args.warmup):
        step()
    # every timed step is also bracketed by a HIP event pair on the stream the step's programs run on (torch's current stream: the
    # side lanes / detached stream fork from it and are joined into it before the step ends)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()
        loss = step()
        evs[i][1].record()
    sync()
    dt = time.perf_counter() - t0
    ev_ms = sorted(a.elapsed_time(b) for a, b in evs)
    med_ms = ev_ms[len(ev_ms) // 2] if len(ev_ms) % 2 else 0.5 * (ev_ms[len(ev_ms) // 2 - 1] + ev_ms[len(ev_ms) // 2])
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss.detach())

    # The profiled step runs AFTER the timed region (it is not part of `value`): every record on ONE stream -- with the branch
    # lanes / the detached weight-gradient stream a launch shares the GPU with other kernels and its event-bracketed time would
    # not be the kernel's own duration -- and HIP events on that stream around every launch (mfc_prof_*).
    rows, prof_steps = [], 0
    if not args.no_prof:                           # (every rank runs the two extra steps: they contain collectives when N > 1)
        prof_steps = 1
        L.lib.mfc_set_flag(9, 0)
        step()                                     # one un-profiled serial step first (stream switch)
        torch.cuda.synchronize()
        L.lib.mfc_prof_enable(1)
        step()
        torch.cuda.synchronize()
        L.lib.mfc_prof_enable(0)
        L.lib.mfc_set_flag(9, 0 if args.serial else 3)
        rows = L.prof_collect()

    # ---- the same step with fp16 storage (adaptive loss scale), reported BESIDE the bf16 headline: 16-bit training fidelity is set by the
    #      forward tensors' rounding (tests/fidelity_probe.py) and fp16 keeps three more mantissa bits there
    plan = next(iter(getattr(model, "_plans", {}).values()), None)
    fused_min = records = None
    if plan is not None and not args.fwd_only:
        # every convolution input read once and every output written once in the forward pass, x3 for training
        # (one read of each saved activation and one write + read of each activation gradient), + Adam's 28 B per parameter
        fused_min = 3 * sum(o[1].t.nbytes + o[2].nbytes for o in plan.ops if o[0] == "conv") + 28 * model._np
        records = (len(plan.fwd_prog) + len(plan.bwd_prog)) if hasattr(plan, "bwd_prog") else None
    del plan
    fp16_side = None
    if (world == 1 and args.dtype == "bf16" and not args.fwd_only and args.fp16_line and not args.single and not args.serial
            and not os.environ.get("MFC_SKIP_KINDS")):
        del model, opt
        torch.cuda.empty_cache()
        m16 = cls(num_classes=nc, num_frames=T, pretrained=False, width=args.width, compute_dtype="fp16",
                  optflow_inputs=args.optflow, depth_inputs=args.depth)
        if args.weights == "hash":
            fill_hashed(m16)
        m16 = m16.to(device).train()
        o16 = mfc.FlatAdam(m16, lr=1e-4)

        def step16():
            return mfc.train_step(m16, o16, frames, mask, optflow=flow, depth=depth)
        for _ in range(3):
            step16()
        n16 = max(5, min(args.steps, 20))
        e16 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n16)]
        torch.cuda.synchronize()
        t16 = time.perf_counter()
        for a_, b_ in e16:
            a_.record(); step16(); b_.record()
        torch.cuda.synchronize()
        t16 = time.perf_counter() - t16
        ms16 = sorted(a_.elapsed_time(b_) for a_, b_ in e16)
        fp16_side = {"value": round(B * T * n16 / t16, 2), "unit": "frames/s", "ms_per_step": round(t16 / n16 * 1e3, 3), "step_ms_median": round(ms16[len(ms16) // 2], 3),
                     "steps": n16, "loss_scale": float(m16.loss_scaler.scale) if getattr(m16, "loss_scaler", None) else None,
                     "skipped_steps": o16.skipped_steps(),
                     "why": "same kernels with IEEE-half storage and an adaptive loss scale; the fidelity of a 16-bit step is set by the rounding of the FORWARD "
                            "tensors (profiles/r03_fidelity_probe.txt), where fp16 keeps 3 more bits than bf16"}
        del m16, o16

    # ---- CPU baseline leg (rank 0, N = 1): the oracle timed on the host cores; it also evaluates the ORACLE'S side of `fp32_parity`
    #      (eval-mode logits of one hash-generated clip) -- the only place bench.py touches oracle/
    cpu, parity_ref = None, None
    want_parity = (world == 1 and rank == 0 and not args.no_cpu_baseline and not args.no_fp32_parity and args.dtype == "bf16" and not args.fwd_only
                   and not args.single and not args.serial and not args.basic and not os.environ.get("MFC_SKIP_KINDS") and args.weights == "hash")
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu, parity_ref = cpu_baseline(args.width, T, H, W, parity_clip=want_parity, optflow=args.optflow, depth=args.depth)
    fp32_parity = None
    if want_parity and parity_ref is not None:
        # the reference computes in fp32; north_star's 1e-3 logit bound is stated for that mode.  Its throughput on the same step
        # and its distance from the CPU oracle on one clip ride beside the bf16 headline, so a driver record carries a number at the reference's precision
        try:
            del model, opt
        except NameError:
            pass
        torch.cuda.empty_cache()
        m32 = cls(num_classes=nc, num_frames=T, pretrained=False, width=args.width, compute_dtype="fp32", optflow_inputs=args.optflow, depth_inputs=args.depth)
        fill_hashed(m32)
        m32 = m32.to(device).train()
        o32 = mfc.FlatAdam(m32, lr=1e-4)
        for _ in range(2):
            mfc.train_step(m32, o32, frames, mask, optflow=flow, depth=depth)
        n32 = 3
        torch.cuda.synchronize()
        t32 = time.perf_counter()
        for _ in range(n32):
            mfc.train_step(m32, o32, frames, mask, optflow=flow, depth=depth)
        torch.cuda.synchronize()
        t32 = time.perf_counter() - t32
        fill_hashed(m32)                                        # (the timed steps moved the weights: back to the fixture state for the parity clip)
        m32.eval()
        pf, pfl, pd = parity_ref["inputs"]
        with torch.no_grad():
            yp = m32([f.to(device) for f in pf], optflow=[f.to(device) for f in pfl] if pfl else None, depth=[f.to(device) for f in pd] if pd else None)
        err = float((yp.float().cpu() - parity_ref["logits"]).abs().max())
        fp32_parity = {"value": round(B * T * n32 / t32, 2), "unit": "frames/s", "ms_per_step": round(t32 / n32 * 1e3, 2), "steps": n32, "dtype": "f32",
                       "logits_max_abs_err_vs_oracle": err, "logits_abs_max": float(parity_ref["logits"].abs().max()), "bound": 1e-3,
                       "clip": f"eval-mode forward of one hash-generated clip (B=1, T={T}, {H}x{W}, key-hash weights) against the CPU oracle's logits",
                       "why": "the reference computes in fp32 and north_star's 1e-3 bound is stated for it: the same step in the fp32 parity mode "
                              "(v_mfma_f32_16x16x4_f32: exact fp32 products, 1/16 of the bf16 MFMA rate) next to the bf16 headline"}
        del m32, o32

    extra = ("+depth" if args.depth else "") + ("+optflow" if args.optflow else "")
    cfg_label = ("per-GPU share of BASELINE.json configs[3]" if (T, H, W, B, extra) == (3, 480, 640, 4, "+depth+optflow") else
                 "BASELINE.json configs[1]" if (args.single, args.fwd_only, H, W, B) == (True, True, 480, 640, 8) else
                 "custom inputs" if (extra or args.single or args.fwd_only) else
                 "BASELINE.json configs[2]" if (T, H, W, B) == (3, 480, 640, 8) else
                 "per-GPU share of BASELINE.json configs[4]" if (T, H, W, B, args.width) == (5, 720, 960, 8, 48) else "custom size")
    if rank == 0:
        roof = roof_conv = roof_top = None
        step_ms = dt / args.steps * 1e3
        timed_rows = [r for r in rows if r["bytes"] > 0 or r["flops"] > 0]
        if timed_rows:
            peak = PEAK_TFLOPS[args.dtype]
            tot_ms = sum(r["ms"] for r in rows)
            # ---- per-TEMPLATE families (every instantiation of one __global__ template is one kernel)
            fams = {}
            for r in rows:
                f = fams.setdefault(r["name"].split("<")[0], {"name": r["name"].split("<")[0], "ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "members": []})
                f["ms"] += r["ms"]; f["flops"] += r["flops"]; f["bytes"] += r["bytes"]; f["launches"] += r["launches"]
                f["members"].append((r["name"], r["launches"]))
            dom = max((f for f in fams.values() if f["bytes"] > 0 or f["flops"] > 0), key=lambda f: f["ms"])
            roof = roofline_of(dom, prof_steps, args.dtype, pmc_traffic(dom["members"], args))
            roof["kernel"] = dom["name"] + "<...> (" + str(len(dom["members"])) + " instantiations of one template)"
            # ---- the whole step against the roofs
            roof_ms = sum(max(r["bytes"] / (PEAK_HBM_GBPS * 1e9), r["flops"] / (peak * 1e12)) for r in rows) * 1e3 / prof_steps
            flops_step = sum(r["flops"] for r in rows) / prof_steps
            bytes_step = sum(r["bytes"] for r in rows) / prof_steps
            ctr_bytes, ctr_src = pmc_traffic(None, args)
            roof.update({
                "step_frac": round(roof_ms / step_ms, 4),
                "step": {"ms": round(step_ms, 3), "roof_ms": round(roof_ms, 3),
                         "roof_ms_is": "sum over every launch of a step of max(algorithmic bytes / 8 TB/s, FLOPs / dense MFMA peak)",
                         "mfma_tflops": round(flops_step / (step_ms * 1e-3) / 1e12, 1), "mfma_frac": round(flops_step / (step_ms * 1e-3) / 1e12 / peak, 4),
                         "algorithmic_gb": round(bytes_step / 1e9, 2), "hbm_frac_algorithmic": round(bytes_step / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         "counter_gb": round(ctr_bytes / 1e9, 2) if ctr_bytes else None,
                         "hbm_frac_counters": round(ctr_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if ctr_bytes else None,
                         "counter_source": ctr_src,
                         "fused_minimum_gb": round(fused_min / 1e9, 2) if fused_min else None,
                         "counters_over_fused_minimum": round(ctr_bytes / fused_min, 2) if (ctr_bytes and fused_min) else None,
                         "serial_step_ms": round(tot_ms / prof_steps, 3), "records_per_step": records},
                "share_of_profiled_step": round(dom["ms"] / tot_ms, 4), "profiled_step_ms": round(tot_ms / prof_steps, 3),
                "profiled_steps": prof_steps, "profiled_step_streams": "serial, after the timed region",
                "families": {f["name"]: {"ms_per_step": round(f["ms"] / prof_steps, 3), "launches": f["launches"] // prof_steps,
                                         "tflops": round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 2) if f["ms"] and f["flops"] else None,
                                         "gbps": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1) if f["ms"] and f["bytes"] else None,
                                         "mfma_frac": round(f["flops"] / (f["ms"] * 1e-3) / 1e12 / peak, 4) if f["ms"] and f["flops"] else None,
                                         "hbm_frac": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if f["ms"] and f["bytes"] else None}
                             for f in sorted(fams.values(), key=lambda f: -f["ms"])[:14]},
                "top5": [{"kernel": r["name"], "ms_per_step": round(r["ms"] / prof_steps, 3), "launches": r["launches"] // prof_steps}
                         for r in rows[:5]]})
            top = timed_rows[0]                                                   # rows come sorted by total time
            roof_top = roofline_of(top, prof_steps, args.dtype, pmc_traffic(top["name"], args))
            convs = [r for r in timed_rows if r["name"].startswith(("conv_igemm_kernel", "conv3x3_ring_kernel"))]
            if convs:
                roof_conv = roofline_of(convs[0], prof_steps, args.dtype, pmc_traffic(convs[0]["name"], args))
        out = {"metric": f"frames/sec ({H}x{W}, T={T}, HRNet {'single-frame' if args.single else 'MFCNet'}) {'fwd' if args.fwd_only else 'fwd+bwd'}", "value": round(world * B * T * args.steps / dt, 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3),
               "step_ms": {"median": round(med_ms, 3), "min": round(ev_ms[0], 3), "max": round(ev_ms[-1], 3), "n": len(ev_ms),
                           "how": "HIP event pair around every timed step, on the step's stream (this rank)"},
               "value_median": round(world * B * T / (med_ms * 1e-3), 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{'single-frame HRNet' if args.single else 'MFCNet'}{'-Basic' if args.basic else ''} T={T} {'RGB' + extra if extra else 'RGB-only'} (HRNet-w{args.width} base), {H}x{W}, batch={B}/GPU, {'eval forward only' if args.fwd_only else 'fwd+bwd+Adam'} "
                                      f"({cfg_label})", "width": args.width, "global_batch": world * B,
                          "frames_per_clip": T, "parallelism": f"dp{world}", "ranks": world,
                          "backend": (args.backend if world > 1 else None), "launcher": os.environ.get("MFC_BENCH_LAUNCHER", "external" if external else "bench.py"),
                          "devices": [f"{h}:cuda{d}" for h, d in devs], "rccl_version": rccl_version(torch) if dist is not None and args.backend == "nccl" else None,
                          "streams": "serial" if args.serial else "branch lanes + detached wgrad",
                          "exchange": (None if world == 1 and not args.force_dist else ("reduce-scatter + sharded Adam + all-gather" if args.sharded_adam else "bucketed all-reduce overlapped with the backward")), "final_loss": round(final_loss, 5)},
               "roofline": roof, "roofline_top": roof_top, "roofline_conv": roof_conv}
        if fp16_side is not None:
            out["fp16_storage"] = fp16_side
        if fp32_parity is not None:
            out["fp32_parity"] = fp32_parity
        out["config"]["weights"] = "key-hash generator (SURVEY.md 8(c)/(d): every state_dict entry a pure function of its key; the golden fixtures' model)" \
            if args.weights == "hash" else "torch default initialisation, manual_seed(1234)"
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
