"""Timeline of the last training step in a rocprofv3 kernel trace (rocpd sqlite) of the multi-stream bench: per stream (queue) the busy
time and its idle gaps, where the chain ends vs where the detached weight-gradient stream ends.
    python tools/timeline.py gpurun_out/prof_round/lanes/lanes_results.db"""
import sqlite3, sys, re
from collections import defaultdict
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from summarize_profile import short
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
# a step starts with the forward's first record: nchw_to_nhwc of frame 0 follows the stats memset; find adam_kernel launches as step ends
adam = [i for i, r in enumerate(rows) if short(r[0]).startswith("adam_kernel")]
# two adam launches per step (two lr groups): step boundaries after every second one
ends = adam[1::2]
if len(ends) < 2:
    print("not enough steps"); sys.exit(0)
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"last step: {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms")
by = defaultdict(list)
for n, s, e, q, st in step:
    by[(q, st)].append((s, e, short(n)))
for key, ks in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, _ in ks)
    first, last = ks[0][0], max(e for _, e, _ in ks)
    fam = defaultdict(float)
    for s, e, n in ks:
        fam[re.sub(r"<.*", "", n)] += (e - s) / 1e6
    top = ", ".join(f"{k} {v:.2f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:5])
    print(f"queue {key[0]} stream {key[1]}: {len(ks):5d} kernels, busy {busy / 1e6:7.3f} ms, active {(first - t0) / 1e6:7.3f} .. {(last - t0) / 1e6:7.3f} ms | {top}")
# where does the forward end (loss kernels)?
for s, e, n in sorted((s, e, short(n)) for n, s, e, q, st in step):
    if n.startswith("loss_fwd") or n.startswith("loss_bwd") or n.startswith("unpack_wgrad") or n.startswith("adam"):
        print(f"  {(s - t0) / 1e6:8.3f} ms  {n}  ({(e - s) / 1e3:.1f} us)")
# concurrency histogram: fraction of the step during which k kernels are running
ev = []
for n, s, e, q, st in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, t0, defaultdict(float)
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels running concurrently: " + ", ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(hist.items())))
# per stream: busy fraction per millisecond of the step (which stream the tail of the step is waiting for)
nb = int((t1 - t0) / 1e6) + 1
print("busy % per ms of the step, by stream (rows) -- columns are ms 0.." + str(nb - 1))
for key, ks in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    occ = [0.0] * nb
    for s, e, _ in ks:
        a_, b_ = (s - t0) / 1e6, (e - t0) / 1e6
        i = int(a_)
        while i < nb and i < b_:
            occ[i] += min(b_, i + 1) - max(a_, i)
            i += 1
    print(f"  stream {key[1]:3d}: " + " ".join(f"{int(100 * min(o, 1.0)):3d}" for o in occ))
