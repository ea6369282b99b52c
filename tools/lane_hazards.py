"""Static race check of a planned program's LANES (CPU only, works on a dry plan).

Records tagged with a lane >= 1 run on parallel HIP streams between two lane-0 records (a fork ... join section, include/mfcnet_hip.h
`mfc_op.lane`); inside a section only same-lane records are ordered.  This walks the forward and backward record lists of a plan, derives
what every record reads and writes (tensor base pointer + channel range, from its descriptor) and reports every pair of records of ONE section
on DIFFERENT lanes where one writes what the other reads or writes.  Detached records (weight gradients, unpack) are checked against the rule
they rely on instead: nothing after them in the program writes what they read.

    python tools/lane_hazards.py [width] [B] [H] [W]        # prints the sections and any hazard; exit code 1 if there is one
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mfcnet-tracker_amd"))


def _v(view, C=None):
    """(ptr, c_lo, c_hi) of a descriptor view; None for an absent one"""
    if not view.ptr:
        return None
    return (int(view.ptr), int(view.c_off), int(view.c_off) + int(C if C is not None else view.Cp))


def _p(ptr):
    return (int(ptr), 0, 1 << 30) if ptr else None


def accesses(kind, d, L):
    """-> (reads, writes): lists of (ptr, c_lo, c_hi)"""
    R, W = [], []
    if kind == L.OP_CONV:
        R += [_p(d.inp), _p(d.in_coef), _p(d.acc_src), _p(d.bn_y), _p(d.bn_bits), _p(d.bn_coef), _p(d.bias), _p(d.wp)]
        W += [_p(d.out), _p(d.out_stats)]
        if d.accumulate and not d.acc_src:
            R.append(_p(d.out))
    elif kind == L.OP_BNFIN:
        R += [_p(d.stats), _p(d.gamma), _p(d.beta)]
        W += [_p(d.coef), _p(d.running_mean), _p(d.running_var)]
    elif kind == L.OP_COMBINE:
        for i in range(d.nsrc):
            R += [_v(d.src[i], d.C), _p(d.src[i].coef)]
        W += [_v(d.out, d.C), _p(d.maskbits)]
    elif kind in (L.OP_BNBWD_REDUCE, L.OP_BNBWD_APPLY):
        R += [_v(d.g, d.C), _v(d.y, d.C), _p(d.y.coef)]
        if d.mask_mode in (1, 3):
            R.append(_v(d.mask, d.C))
        if kind == L.OP_BNBWD_REDUCE:
            W += [_p(d.bstats), _v(d.dy, d.C)]
            if d.accumulate:
                R.append(_v(d.dy, d.C))
        else:
            R += [_p(d.bstats), _p(d.bcoef)]
            W += [_v(d.dy, d.C), _p(d.fin_dgamma), _p(d.fin_dbeta)]
    elif kind == L.OP_BNBWD_FIN:
        R.append(_p(d.bstats))
        W += [_p(d.bcoef), _p(d.dgamma), _p(d.dbeta)]
    elif kind == L.OP_MASK_ADD:
        R += [_v(d.g, d.C)]
        if d.mask_mode in (1, 3):
            R.append(_v(d.mask, d.C))
        if d.accumulate:
            R.append(_v(d.dst, d.C))
        W += [_v(d.dst, d.C), _p(d.scratch)]
    elif kind == L.OP_WGRAD:
        R += [_p(d.x), _p(d.dy), _p(d.in_coef)]
        W.append(_p(d.dwp))
    elif kind == L.OP_BIAS_GRAD:
        R.append(_p(d.a))
        W.append(_p(d.b))
    return [a for a in R if a], [a for a in W if a]


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def check(recs, L, name="program"):
    """recs: list of (kind, desc, lane).  Returns (sections, hazards) -- hazards as printable strings."""
    hazards, sections = [], []
    cur = []
    acc = [None] * len(recs)
    for i, (kind, d, lane) in enumerate(recs):
        if kind == L.OP_WGRAD_BATCH:
            n = d.i[0]
            arr = (L.WgradDesc * n).from_address(d.a)
            rr, ww = [], []
            for k in range(n):
                r1, w1 = accesses(L.OP_WGRAD, arr[k], L)
                rr += r1; ww += w1
            acc[i] = (rr, ww)
        else:
            acc[i] = accesses(kind, d, L)
    for i, (kind, d, lane) in enumerate(recs):
        if lane & L.LANE_ASYNC:
            continue                                   # detached: not a member of the section (checked below)
        if (lane & 0xff) == 0:
            if cur:
                sections.append(cur); cur = []
            continue
        cur.append(i)
    if cur:
        sections.append(cur)
    for sec in sections:
        for x in range(len(sec)):
            i = sec[x]
            Ri, Wi = acc[i]
            for y in range(x + 1, len(sec)):
                j = sec[y]
                if (recs[i][2] & 0xff) == (recs[j][2] & 0xff):
                    continue
                Rj, Wj = acc[j]
                for a in Wi:
                    for b in Rj + Wj:
                        if _overlap(a, b):
                            hazards.append(f"{name}: record {i} (kind {recs[i][0]}, lane {recs[i][2] & 0xff}) writes what record {j} (kind {recs[j][0]}, lane {recs[j][2] & 0xff}) "
                                           f"{'reads' if b in Rj else 'writes'}: ptr {a[0]:#x} channels [{max(a[1], b[1])}, {min(a[2], b[2])})")
                for a in Ri:
                    for b in Wj:
                        if _overlap(a, b):
                            hazards.append(f"{name}: record {j} (kind {recs[j][0]}, lane {recs[j][2] & 0xff}) overwrites what record {i} (kind {recs[i][0]}, lane {recs[i][2] & 0xff}) "
                                           f"reads: ptr {a[0]:#x} channels [{max(a[1], b[1])}, {min(a[2], b[2])})")
    # detached records: what they read must not be written by any later record (they run whenever the detached stream gets to them)
    for i, (kind, d, lane) in enumerate(recs):
        if not (lane & L.LANE_ASYNC):
            continue
        Ri, _ = acc[i]
        for j in range(i + 1, len(recs)):
            for b in acc[j][1]:
                for a in Ri:
                    if _overlap(a, b):
                        hazards.append(f"{name}: detached record {i} (kind {kind}) reads ptr {a[0]:#x}, which the later record {j} (kind {recs[j][0]}) writes")
    return sections, hazards


def check_plan(pl, L):
    out = {}
    hz = []
    for name, recs in (("forward", list(pl.fwd)), ("backward", list(pl.bwd))):
        secs, h = check(recs, L, name)
        out[name] = secs
        hz += h
    return out, hz


def main():
    import torch
    import mfcnet_amd as mfc
    from mfcnet_amd import _lib as L
    from mfcnet_amd.plan import Plan
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    m = mfc.HRNetMultiLarge(num_classes=5, num_frames=3, pretrained=False, width=width, compute_dtype="bf16").train()
    pl = Plan(m, B, H, W, False, False, True, True, True, torch.device("cpu"), dry=True)
    secs, hz = check_plan(pl, L)
    for name, ss in secs.items():
        print(f"{name}: {len(ss)} parallel sections, {sum(len(s) for s in ss)} lane records, longest {max(len(s) for s in ss)}")
    for h in hz[:40]:
        print("HAZARD", h)
    print(f"{len(hz)} hazards")
    return 1 if hz else 0


if __name__ == "__main__":
    sys.exit(main())
