"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/mfcnet_hip.h declares,
the ctypes mirrors match the C struct sizes, descriptors are validated before any launch, and the host-side
mirror of the reference interface (names, state_dict, errors) behaves like the reference's."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mfcnet_hip.h")


@pytest.fixture(scope="module")
def L():
    from mfcnet_amd import _lib
    return _lib


TUNING_HEADER = os.path.join(ROOT, "include", "mfcnet_hip_tuning.h")


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(mfc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(L):
    names = declared_functions()
    assert len(names) >= 25
    lib = C.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(L.EXPORTS), set(names) ^ set(L.EXPORTS)
    assert b"gfx950" in L.lib.mfc_version()
    # the tuning switches are a separate, private header: exported for tools/, absent from the product interface
    tuning = declared_functions(TUNING_HEADER)
    assert tuning == sorted(L.TUNING_EXPORTS) == ["mfc_set_flag"] and not (set(tuning) & set(names))
    assert all(hasattr(lib, n) for n in tuning)
    # the context API: handles are created without touching a device, validated, destroyed
    h = C.c_void_p()
    assert L.lib.mfc_ctx_create(0, C.byref(h)) == 0 and h.value
    assert L.lib.mfc_ctx_destroy(h) == 0
    assert L.lib.mfc_ctx_create(-1, C.byref(h)) == -1 and L.lib.mfc_ctx_destroy(None) == -1


def test_ctypes_mirrors_match_c_struct_sizes(L):
    structs = {"mfc_op": L.Op, "mfc_conv_desc": L.ConvDesc, "mfc_wgrad_desc": L.WgradDesc, "mfc_pack_job": L.PackJob,
               "mfc_unpack_job": L.UnpackJob, "mfc_bnfin_desc": L.BnFinDesc, "mfc_gnfin_desc": L.GnFinDesc, "mfc_view": L.View, "mfc_combine_desc": L.CombineDesc,
               "mfc_bnbwd_desc": L.BnBwdDesc, "mfc_bnbwdfin_desc": L.BnBwdFinDesc, "mfc_maskadd_desc": L.MaskAddDesc,
               "mfc_headgather_desc": L.HeadDesc, "mfc_loss_desc": L.LossDesc, "mfc_conv_layout": L.ConvLayout,
               "mfc_prof_entry": L.ProfEntry}
    prog = '#include "mfcnet_hip.h"\n#include <stdio.h>\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in structs) + \
        'printf("STAT_REPLICAS %d\\nSTAT_BYTES %zu\\nF16 %d\\nLOSS_ACC_FLOATS %d\\n", MFC_STAT_REPLICAS, sizeof(mfc_stat_t), (int)MFC_F16, MFC_LOSS_ACC_FLOATS);return 0;}'
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "s.c"), os.path.join(td, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    consts = {"STAT_REPLICAS": L.STAT_REPLICAS, "STAT_BYTES": L.STAT_BYTES, "F16": L.F16, "LOSS_ACC_FLOATS": L.LOSS_ACC_FLOATS}      # constants the Python side mirrors
    for line in out.strip().splitlines():
        name, size = line.split()
        if name in consts:
            assert consts.pop(name) == int(size), name
        else:
            assert C.sizeof(structs[name]) == int(size), name
    assert not consts
    assert L.lib.mfc_op_size() == C.sizeof(L.Op)


def test_descriptors_are_validated_before_any_launch(L):
    assert L.lib.mfc_conv2d_fwd(C.byref(L.ConvDesc()), None) == -1
    assert L.lib.mfc_conv2d_wgrad(C.byref(L.WgradDesc()), None) == -1
    assert L.lib.mfc_bn_finalize(C.byref(L.BnFinDesc()), None) == -1
    assert L.lib.mfc_bn_finalize_batch(None, 3, 64, None) == -1
    assert L.lib.mfc_combine_fwd(C.byref(L.CombineDesc()), None) == -1
    assert L.lib.mfc_program_run(None, 3, None) == -1
    d = L.ConvDesc(16, 16, 16, 0, 0, 0, L.BF16, 2, 8, 8, 12, 12, 8, 8, 16, 16, 8, 8, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, 2, 0, 0, 0)
    assert L.lib.mfc_conv2d_fwd(C.byref(d), None) == -1            # channel pitch 12 is not a multiple of 8
    d.Cin_p = 16
    lay = L.conv_layout(d)                                          # geometry query needs no GPU
    assert lay.TA == 3 and lay.NT16 == 16 and lay.bytes > 0 and lay.lds_bytes <= 80 * 1024
    # NULL descriptors are an invalid argument for every convolution entry point, not a crash (round-2 advisor finding)
    lay2 = L.ConvLayout()
    assert L.lib.mfc_conv2d_layout(None, C.byref(lay2)) == -1
    assert L.lib.mfc_conv2d_lds_bytes(None) == -1
    assert L.lib.mfc_conv2d_fwd(None, None) == -1
    # the LDS-DMA ring kernel takes the 32 / 64-channel 3x3 BasicBlock convolutions: its weight image is [tap][granule][cout]
    r = L.ConvDesc(16, 16, 16, 0, 0, 0, L.BF16, 24, 120, 160, 32, 32, 120, 160, 32, 32, 120, 160, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, 8, 0, 0, 0)
    lr = L.conv_layout(r)
    assert (lr.KG, lr.nchunks, lr.Yblocks, lr.NT16, lr.nslots, lr.TAS, lr.fa) == (4, 1, 1, 32, 36, 3, 1) and lr.bytes == 9 * 32 * 32 * 2
    assert 2 * lr.lds_bytes <= 160 * 1024                           # two workgroups per CU
    L.lib.mfc_set_flag(30, 0)
    assert L.conv_layout(r).nslots != 36 or L.conv_layout(r).NT16 != 32 or L.conv_layout(r).fa == 1      # (conv_igemm's own blocking)
    L.lib.mfc_set_flag(30, 1)


def test_model_mirrors_reference_interface():
    import mfcnet_amd as mfc
    from oracle import mfcnet_oracle as O
    args = SimpleNamespace(model_type="HRNetMulti-Large", num_classes=5, num_input_frames=3, pretrained=True,
                           load_wts_base_model=None, add_optflow_inputs=True, add_depth_inputs=True)
    m = mfc.get_multiframe_segmentation_model(args)
    ref_keys = [t[0] for t in O.mfcnet_table("HRNetMulti-Large", 48, 5, 3, True, True)]
    assert list(m.state_dict().keys()) == ref_keys                 # names AND order of the reference state_dict
    sd = O.hashed_state(O.mfcnet_table("HRNetMulti-Large", 48, 5, 3, True, True))
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.base_model.last_layer[3].bias.detach(), sd["base_model.last_layer.3.bias"])
    assert m.base_model.conv1.weight.data_ptr() == m._P.data_ptr()                # parameters are views of ONE arena
    import numpy as np
    n_ref = sum(int(np.prod(sh)) for _, sh, k in O.mfcnet_table('HRNetMulti-Large', 48, 5, 3, True, True) if k in ('conv_w', 'conv_b', 'bn_gamma', 'bn_beta'))
    assert sum(p.numel() for p in m.parameters()) == n_ref
    # the two optimizer groups of scripts/train_multiframe_detection.py:128-151 are two contiguous segments
    seg = m.flat_segments()
    assert seg["base_model"][1] == seg["multiframe_net"][0] and seg["multiframe_net"][1] == m._P.numel()
    m.train(); m.base_model.eval()
    assert m.multiframe_net.training and not m.base_model.training
    args.model_type = "HRNetMulti-Basic"
    mb = mfc.get_multiframe_segmentation_model(args)
    assert "multiframe_net.grid" in mb.state_dict() and mb.state_dict()["multiframe_net.grid"].shape == (1, 2, 576, 720)
    args.model_type = "UNet"
    with pytest.raises(ValueError, match="not recognized"):        # models/__init__.py:86
        mfc.get_multiframe_segmentation_model(args)
    x = [torch.zeros(1, 3, 64, 96) for _ in range(3)]
    with pytest.raises(mfc.MfcError):                               # no CPU fallback: the product path fails loudly
        m(x, optflow=[torch.zeros(1, 2, 64, 96)] * 2, depth=[torch.zeros(1, 1, 64, 96)] * 3)


def test_data_parallel_replication_is_refused_with_a_clear_error():
    """scripts/train_multiframe_detection.py:107-110 wraps the model in nn.DataParallel; on a multi-GPU node that replicates the module
    from one process, which this model cannot do: it must say so (and name the stand-in), not fail inside replicate()."""
    import mfcnet_amd as mfc
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=32)
    with pytest.raises(mfc.MfcError, match="mfcnet_amd.DataParallel"):
        m._replicate_for_data_parallel()
    dp = mfc.DataParallel(m)                      # no process group: a transparent wrapper
    assert dp.module is m and dp.reducer is None and next(iter(dp.state_dict())).startswith("module.")
    assert dp.module.base_model is m.base_model and dp.module.multiframe_net is m.multiframe_net


def test_plan_builds_without_gpu():
    """The whole forward+backward program (addresses, launch geometry, packed-weight layouts) is planned on the host."""
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, optflow_inputs=True, depth_inputs=True, compute_dtype="bf16")
    p = Plan(m, 2, 64, 96, True, True, True, True, True, torch.device("cpu"))
    kinds = [o.kind for o in p.fwd_prog]
    assert kinds.count(L.OP_CONV) == 311 and kinds.count(L.OP_BNFIN) == 309        # 307 + 4 convs
    bk = [o.kind for o in p.bwd_prog]
    assert bk.count(L.OP_WGRAD) == 311 and bk[-1] == L.OP_UNPACK
    # every BatchNorm has an apply record; its reduce is either a record of its own or folded into the data-gradient launch that
    # completes the gradient it reduces (mfc_conv_desc.bn_y)
    fused = sum(1 for o in p.bwd_prog if o.kind == L.OP_CONV and o.u.conv.bn_y)
    assert bk.count(L.OP_BNBWD_APPLY) == 309 and 0 < bk.count(L.OP_BNBWD_REDUCE) < 309 and fused >= 309 - bk.count(L.OP_BNBWD_REDUCE)
    m.fuse_bnbwd_reduce = False
    p0 = Plan(m, 2, 64, 96, True, True, True, True, True, torch.device("cpu"), dry=True)
    assert sum(1 for r in p0.bwd if r[0] == L.OP_BNBWD_REDUCE) == 309


def test_gradient_buckets_partition_the_backward_program():
    """plan.py::_build_grad_buckets (host logic, no GPU): the segments' ranges tile the flat gradient arena exactly once, every
    weight-gradient unpack job sits in the segment of its bucket, and a bucket's unpack comes after the last record that writes
    into it."""
    import torch
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=16, compute_dtype="bf16")
    pl = Plan(m, 2, 64, 96, False, False, True, True, True, torch.device("cpu"))
    ranges = sorted((lo, hi) for _, lo, hi in pl.bwd_segments)
    assert ranges[0][0] == 0 and ranges[-1][1] == m._np
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1))
    nseg_unpack = sum(1 for prog, _, _ in pl.bwd_segments for i in range(len(prog)) if prog[i].kind == L.OP_UNPACK)
    # same records; one unpack per segment there, chunked detached unpacks (every 48 weight gradients, the rest at the end) in the one program
    one = [pl.bwd_prog[i] for i in range(len(pl.bwd_prog))]
    n_one_unpack = sum(1 for o in one if o.kind == L.OP_UNPACK)
    assert sum(len(prog) for prog, _, _ in pl.bwd_segments) - nseg_unpack == len(one) - n_one_unpack
    assert n_one_unpack == -(-sum(1 for o in one if o.kind == L.OP_WGRAD) // 48) and one[-1].kind == L.OP_UNPACK
    assert all((o.lane & L.LANE_ASYNC) for o in one if o.kind == L.OP_UNPACK)       # ... all on the detached stream, behind their launches
    # every parameter's last writer precedes (or is) the end of the segment that owns its bucket
    seg_end, k = [], 0
    for prog, lo, hi in pl.bwd_segments:
        k += len(prog)
        seg_end.append((lo, hi, k))
    G = m._G.data_ptr()
    n_unpack = 0
    for prog, lo, hi in pl.bwd_segments:
        for i in range(len(prog)):
            if prog[i].kind == L.OP_UNPACK:
                n_unpack += prog[i].u.raw.i[0]
    n_bias = sum(1 for o in one if o.kind == L.OP_BIAS_GRAD and o.u.raw.i[3] > 0)      # bias gradients are partial-sum slices too
    assert n_unpack == len(pl.unpack_jobs) == 311 + n_bias and n_bias == 2
    # buckets complete from the end of the arena (the head is first in backward)
    assert pl.bwd_segments[0][2] == m._np



def test_dry_plan_enumerates_the_benchmarked_launches_without_a_gpu():
    """Plan(dry=True) lays the whole step out with placeholder addresses (no device memory): the geometry search of every launch of
    BASELINE configs[2] runs on the host, and tests/test_gpu_plan_kernels.py replays exactly these descriptors on the GPU."""
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=32, compute_dtype="bf16").train()
    pl = Plan(m, 8, 480, 640, False, False, True, True, True, torch.device("cpu"), dry=True)
    convs = [op[3] for op in pl.ops if op[0] == "conv"]
    assert len(convs) == 311                                         # 307 base + 4 head convolutions
    assert all(ci.fwd is not None and ci.wg is not None for ci in convs)
    assert sum(1 for ci in convs if not ci.dgrad) == 1               # only the stem conv (its input are the frames) has no data gradient
    for ci in convs[:20]:
        lay = _lib.conv_layout(ci.fwd)
        assert lay.bytes > 0 and lay.NW in (4, 8)
        assert _lib.wgrad_parts(ci.wg) >= 1
    # frozen per-frame network (the reference's default mode): the backward is the temporal head's only
    for p in m.base_model.parameters():
        p.requires_grad = False
    plf = Plan(m, 8, 480, 640, False, False, False, True, True, torch.device("cpu"), base_frozen=True, dry=True)
    assert len(plf.bwd) < 40 and not any(r[0] == _lib.OP_HEAD_BWD for r in plf.bwd)
    assert len(pl.bwd) > 1000


def test_product_key_hash_generator_equals_the_oracles():
    """mfcnet_amd.synth (bench.py's weights) against the oracle's `hashed_state`, entry by entry: same keys in the same order,
    same bits -- for both model types (with flow / depth inputs: the Basic model's registered grid is kept, not generated) and the single-frame net."""
    import mfcnet_amd as mfc
    from mfcnet_amd.synth import fill_hashed, hashed_state_for
    from oracle import mfcnet_oracle as O
    for cls, mt, T, fl, dp, w in ((mfc.HRNetMultiLarge, "HRNetMulti-Large", 3, False, False, 32), (mfc.HRNetMultiBasic, "HRNetMulti-Basic", 3, True, True, 48)):
        m = cls(num_classes=5, num_frames=T, pretrained=False, width=w, optflow_inputs=fl, depth_inputs=dp)
        a, b = hashed_state_for(m), O.hashed_state(O.mfcnet_table(mt, w, 5, T, fl, dp))
        assert list(a) == list(b)
        assert all(torch.equal(a[k], b[k]) for k in a), [k for k in a if not torch.equal(a[k], b[k])][:3]
    s = mfc.HighResolutionNetHIP(num_classes=5, width=48)
    a, b = hashed_state_for(s), O.hashed_state(O.hrnet_table(48, 5, ""))
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    fill_hashed(s)
    assert torch.equal(s.state_dict()["conv1.weight"], b["conv1.weight"])


def test_committed_pmc_table_says_which_code_it_was_taken_on(tmp_path, monkeypatch):
    """bench.py prices `roofline.traffic` / `step.counter_gb` from the newest committed PMC table: the table carries a hash of the kernel
    sources, headers and planner it was taken on, and the bench line says so -- or says STALE when the sources have moved on."""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(width=32, dtype="bf16", batch=8, frames=3, height=480, width_px=640, depth=False, optflow=False, basic=False,
                              single=False, fwd_only=False)
    val, src = bench.pmc_traffic(None, args)
    assert val and val > 5e10 and "profiles/r" in src
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_w32_pmc_traffic.json")))[-1]
    doc = json.load(open(newest))
    assert ("taken on this code" in src) == (doc.get("code_hash") == bench.code_hash())
    assert ("STALE" in src) != ("taken on this code" in src)
    monkeypatch.setattr(bench, "code_hash", lambda: "0" * 16)
    assert "STALE" in bench.pmc_traffic(None, args)[1]


@pytest.mark.parametrize("H,W,width", [(480, 640, 32), (960, 1280, 32), (480, 640, 48), (720, 960, 48)],
                         ids=["w32-480x640", "w32-960x1280", "w48-480x640", "w48-720x960"])
def test_every_planned_launch_reads_the_weight_image_that_was_packed_for_it(H, W, width):
    """The packed weight image of a convolution launch follows the launch geometry mfc_conv2d_layout chose when the plan was built; the planner fills
    `accumulate` / `acc_src` / `bn_y` in later.  Whatever those fields become, the launch the library picks for the FINAL descriptor must still be the
    one the image was packed for (a kernel choice that depended on a late field would silently read another kernel's layout): checked for every
    convolution record of the forward and backward programs on a dry plan -- also at a size where the 64-channel branch is past the pixel count
    from which data gradients that promise never to accumulate (MFC_CONV_NEVER_ACC) take the unfused ring launch."""
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.plan import Plan
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=width, compute_dtype="bf16").train()
    pl = Plan(m, 8, H, W, False, False, True, True, True, torch.device("cpu"), dry=True)
    jobs = {}
    for j in pl.pack_jobs:
        jobs.setdefault(j["dst"], j)
    keys = ("KG", "nchunks", "NT16", "Yblocks", "nslots", "TAS")
    n = ring64 = 0
    for rec in list(pl.fwd) + list(pl.bwd):
        if rec[0] != L.OP_CONV:
            continue
        d = rec[1]
        lay = L.conv_layout(d)
        job = jobs[d.wp]
        now = L.pack_job_fields(lay)
        assert all(job[k] == now[k] for k in keys), (d.Cin, d.Cout, d.TA, d.Hin, d.accumulate, bool(d.acc_src), bool(d.bn_y), {k: (job[k], now[k]) for k in keys})
        n += 1
        if d.Cin == 64 and d.Cout == 64 and d.TA == 3 and (d.flags & L.CONV_NEVER_ACC) and lay.fa == 0:
            assert not d.accumulate and not d.acc_src and not d.bn_y
            ring64 += 1
    assert n > 500
    # layer1's four conv2 data gradients (every width); W32 also the 32 conv2's of its 64-channel branch (N * H * W >= 100 000 pixels at both sizes)
    assert ring64 == (4 + 32 if width == 32 else 4)


@pytest.mark.parametrize("width,basic,aux,one_join", [(32, False, False, True), (32, False, False, False), (48, False, False, True), (32, False, True, True), (32, True, True, True)],
                         ids=["w32", "w32-round3-fuse-order", "w48", "w32-flow-depth", "w32-basic-flow-depth"])
def test_lanes_of_the_planned_programs_are_race_free(width, basic, aux, one_join):
    """tools/lane_hazards.py: inside a fork ... join section of a program only same-lane records are ordered.  From the descriptors of a dry plan:
    no record of a section writes a tensor (base pointer + channel range) that a record on ANOTHER lane of the same section reads or writes, and
    nothing after a detached record (weight gradient) writes what it reads.  Run for the benchmarked plans, both emission orders of the fuse stage,
    and -- so that a checker that finds nothing is known to be able to find something -- for a plan whose one-join-per-module record is dropped."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lane_hazards as H
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd import plan as P
    cls = mfc.HRNetMultiBasic if basic else mfc.HRNetMultiLarge
    m = cls(num_classes=5, num_frames=3, pretrained=False, width=width, compute_dtype="bf16", optflow_inputs=aux, depth_inputs=aux).train()
    m.fuse_one_join = one_join
    pl = P.Plan(m, 4, 480, 640, aux, aux, True, True, True, torch.device("cpu"), dry=True)
    secs, hz = H.check_plan(pl, L)
    assert len(secs["forward"]) >= 8 and len(secs["backward"]) >= 8
    assert not hz, hz[:5]
    if width == 32 and one_join and not aux:
        orig = P.Plan.join
        try:
            def no_join(self):
                self.cur_lane = 0
                self.ops.append(("nojoin",))          # (no record: the fuse sums then share a section with the paths they read)
            P.Plan.join = no_join
            bad = P.Plan(m, 4, 480, 640, False, False, True, True, True, torch.device("cpu"), dry=True)
            assert len(H.check_plan(bad, L)[1]) > 50
        finally:
            P.Plan.join = orig


def test_never_acc_promise_decides_the_launch_before_the_weights_are_packed():
    """MFC_CONV_NEVER_ACC (include/mfcnet_hip.h): a 64-channel 3x3 data gradient that asks for a fusable launch gets conv_igemm's (fa = 1) -- unless it
    promises never to accumulate AND the image is large, where the plain ring launch (no fusions, its own weight layout) is chosen.  The choice must be a
    function of what is known when mfc_conv2d_layout is asked, i.e. it must not change when `accumulate` is filled in afterwards."""
    from mfcnet_amd import _lib as L

    def desc(flags, N=24, H=60, W=80, acc=0):
        d = L.ConvDesc(16, 16, 16, 0, 0, 0, L.BF16, N, H, W, 64, 64, H, W, 64, 64, H, W, 3, 3, -1, -1, 1, 1, 1, 0, 0, 0, N // 3, acc, 0, 0)
        d.flags = flags
        return d
    geo = lambda lay: (lay.fa, lay.NW, lay.MT, lay.TH, lay.TW, lay.grid) + tuple(sorted(L.pack_job_fields(lay).items()))
    ring = L.conv_layout(desc(L.CONV_WANT_FA | L.CONV_NEVER_ACC))
    assert (ring.fa, ring.NW, ring.TH, ring.TW, ring.nchunks) == (0, 4, 8, 16, 1)                # the ring launch: 8 x 16 tiles, all weights resident, no fusions offered
    fused = L.conv_layout(desc(L.CONV_WANT_FA))
    assert fused.fa == 1 and geo(fused) != geo(ring)                                             # conv_igemm's fusable launch
    later = L.conv_layout(desc(L.CONV_WANT_FA, acc=1))                                           # (what the planner fills in after packing)
    assert geo(later) == geo(fused)
    small = L.conv_layout(desc(L.CONV_WANT_FA | L.CONV_NEVER_ACC, N=3))                          # 14 400 pixels: below the threshold
    assert small.fa == 1


def test_launch_choice_never_depends_on_fields_the_planner_fills_in_later():
    """Property check of mfc_conv2d_layout over every residual-block shape and 400 pseudo-random data-gradient descriptors (1x1 / 3x3, stride 1 and the stride-2 forms, 16-bit, channel
    counts of HRNet-W32 / W48 / the head, image sizes from 15x20 to 120x160, with and without MFC_CONV_WANT_FA / MFC_CONV_NEVER_ACC): the packed
    weight layout of the launch chosen for the descriptor as the planner first asks about it must be that of the launch chosen after `accumulate`, and -- where fa = 1 was
    reported -- `acc_src`, `bn_y` / `out_stats` have been filled in.  (The kernel families dispatch on eligibility predicates; one that looked at
    such a field would make a launch read weights packed for another kernel.)"""
    import random
    from mfcnet_amd import _lib as L
    rnd = random.Random(1234)
    chans = [5, 15, 16, 32, 48, 64, 96, 128, 192, 256, 384, 480]
    sizes = [(15, 20), (30, 40), (60, 80), (120, 160), (17, 23)]
    # what must not move: the packed weight image (its blocking and size); and a launch that was offered the fusions must still take them
    geo = lambda lay: (lay.bytes,) + tuple(sorted(L.pack_job_fields(lay).items()))
    checked = fusable = 0
    # every residual-block shape (equal channels, 3x3 / stride 1: the ring kernel's territory) explicitly, then 400 random ones
    fixed = [(c, c, hw, n, 3, "s1", fl) for c in chans for hw in sizes for n in (3, 24) for fl in (0, L.CONV_WANT_FA, L.CONV_WANT_FA | L.CONV_NEVER_ACC)]
    for it in range(len(fixed) + 400):
        if it < len(fixed):
            cin, cout, (H, W), N, k, form, fixed_flags = fixed[it]
        else:
            fixed_flags = None
            cin, cout = rnd.choice(chans), rnd.choice(chans)
            H, W = rnd.choice(sizes)
            N = rnd.choice([3, 6, 12, 24])
            k = rnd.choice([1, 3])
            form = rnd.choice(["s1", "s1", "s2_class", "s2_all"]) if k == 3 else "s1"
        cinp, coutp = -(-cin // 8) * 8, -(-cout // 8) * 8
        pad = k // 2
        if form == "s1":
            args = (N, H, W, cinp, cin, H, W, coutp, cout, H, W, k, k, -(k - 1 - pad), -(k - 1 - pad), 1, 1, 1, 0, 0)
        elif form == "s2_all":
            Ho, Wo = 2 * H, 2 * W
            args = (N, H, W, cinp, cin, Ho, Wo, coutp, cout, (Ho + 1) // 2, (Wo + 1) // 2, 2, 2, 0, 0, 1, 2, 2, 0, 0)
        else:
            Ho, Wo = 2 * H, 2 * W
            ph, pw = rnd.randrange(2), rnd.randrange(2)
            ta = 2 if ph == 1 else 1
            tb = 2 if pw == 1 else 1
            args = (N, H, W, cinp, cin, Ho, Wo, coutp, cout, (Ho - ph + 1) // 2, (Wo - pw + 1) // 2, ta, tb, 0, 0, 1, 2, 2, ph, pw)
        flags = rnd.choice([0, L.CONV_WANT_FA, L.CONV_WANT_FA, L.CONV_WANT_FA | L.CONV_NEVER_ACC]) | (L.CONV_S2_CLASSES if form == "s2_all" else 0)
        if fixed_flags is not None:
            flags = fixed_flags

        def make(**kw):
            d = L.ConvDesc(16, 16, 16, 0, 0, 0, L.BF16, *args, 0, N // 3, 0, 0, 0)
            d.flags = flags
            for key, val in kw.items():
                setattr(d, key, val)
            return d
        try:
            base = L.conv_layout(make())
        except L.MfcError:
            continue
        checked += 1
        variants = []
        if not (flags & L.CONV_NEVER_ACC):
            variants.append(dict(accumulate=1))
        if base.fa == 1:
            fusable += 1
            variants.append(dict(bn_y=16, bn_coef=16, out_stats=16, bn_mask_mode=2))
            variants.append(dict(bn_y=16, bn_coef=16, out_stats=16, bn_mask_mode=3, bn_bits=16))
            if not (flags & L.CONV_NEVER_ACC):
                variants.append(dict(accumulate=1, acc_src=16))
                variants.append(dict(accumulate=1, acc_src=16, bn_y=16, bn_coef=16, out_stats=16, bn_mask_mode=3, bn_bits=16))
        for kw in variants:
            lay = L.conv_layout(make(**kw))
            assert geo(lay) == geo(base), (cin, cout, k, form, H, W, N, flags, kw)
            if kw.get("bn_y") or kw.get("acc_src"):
                assert lay.fa == 1, (cin, cout, k, form, H, W, N, flags, kw)
    assert checked > 600 and fusable > 150, (checked, fusable)
