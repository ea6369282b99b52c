"""Condense the rocprofv3 (rocpd sqlite) outputs of tools/profile_round.sh into small tracked files under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_round r01_d_w32

Writes profiles/<tag>_kernel_stats_{serial,lanes}.csv (per-kernel calls / total / average / min / max duration, the
`--stats` table) and profiles/<tag>_pmc_traffic.json (per-kernel average FETCH_SIZE / WRITE_SIZE per dispatch from the two
separate PMC passes, with the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 64 B per 128-B request on
wide coalesced reads -> doubled; WRITE_SIZE taken as is; both counters are in KiB-sized units of 1024 B).
"""
import csv, json, os, re, sqlite3, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, duration from kernels").fetchall()
    agg = defaultdict(list)
    for n, d in rows:
        agg[short(n)].append(d)
    tot = sum(sum(v) for v in agg.values())
    out = []
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append(dict(kernel=n, calls=len(v), total_ns=sum(v), avg_ns=round(sum(v) / len(v), 1), min_ns=min(v), max_ns=max(v),
                        percent=round(100.0 * sum(v) / tot, 2)))
    return out


def pmc_avg(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = defaultdict(list)
    for n, v in rows:
        agg[short(n)].append(v)
    return {n: (sum(v) / len(v), len(v)) for n, v in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    for mode in ("serial", "lanes"):
        p = os.path.join(src, mode, f"{mode}_results.db")
        if not os.path.exists(p):
            continue
        st = kernel_stats(p)
        with open(os.path.join(root, "profiles", f"{tag}_kernel_stats_{mode}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(st[0].keys()))
            w.writeheader(); w.writerows(st)
        print(mode, "top kernels:")
        for r in st[:8]:
            print(f"  {r['percent']:6.2f}%  n={r['calls']:5d}  avg {r['avg_ns'] / 1e3:8.1f} us  {r['kernel'][:90]}")
    pf, pw = os.path.join(src, "pmc_fetch", "fetch_results.db"), os.path.join(src, "pmc_write", "write_results.db")
    if os.path.exists(pf) and os.path.exists(pw):
        fe, wr = pmc_avg(pf, "FETCH_SIZE"), pmc_avg(pw, "WRITE_SIZE")
        out = {"note": "per-dispatch averages over one serial training step; FETCH_SIZE doubled (gfx950 wide-read correction), units of 1024 B",
               "kernels": {}}
        for n in fe:
            f_kb, calls = fe[n]
            w_kb = wr.get(n, (0.0, 0))[0]
            out["kernels"][n] = {"dispatches": calls, "fetch_size_raw_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                                 "hbm_bytes_per_launch": round((2.0 * f_kb + w_kb) * 1024.0)}
        with open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        for n in sorted(out["kernels"], key=lambda k: -out["kernels"][k]["hbm_bytes_per_launch"] * out["kernels"][k]["dispatches"])[:8]:
            k = out["kernels"][n]
            print(f"  pmc {k['dispatches']:5d} x {k['hbm_bytes_per_launch'] / 1e6:9.2f} MB  {n[:90]}")


if __name__ == "__main__":
    main()
