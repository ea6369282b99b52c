"""Condense the rocprofv3 (rocpd sqlite) outputs of tools/profile_round.sh into small tracked files under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_round r01_d_w32

Writes profiles/<tag>_kernel_stats_{serial,lanes}.csv (per-kernel calls / total / average / min / max duration, the
`--stats` table) and profiles/<tag>_pmc_traffic.json (per-kernel average FETCH_SIZE / WRITE_SIZE per dispatch from the two
separate PMC passes, with the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 64 B per 128-B request on
wide coalesced reads -> doubled; WRITE_SIZE taken as is; both counters are in KiB-sized units of 1024 B).
"""
import csv, json, os, re, sqlite3, sys
from collections import defaultdict


_demangled = {}


def _demangle(name):
    """Itanium names of this library's kernel templates (rocprofv3 prints some of them mangled; binutils' c++filt does not know the
    __bf16 code DF16b): _Z<len><name>I<args>E... with args in {DF16b, f, Li<n>E, Lb<0|1>E}."""
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16b", i):
            args.append("__bf16"); i += 5
        elif rest.startswith("DF16_", i):
            args.append("_Float16"); i += 5
        elif rest[i] == "f":
            args.append("float"); i += 1
        elif rest.startswith("Lb", i):
            args.append("true" if rest[i + 2] == "1" else "false"); i += 4
        elif rest.startswith("Li", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append("-" + v[1:] if v.startswith("n") else v); i = j + 1
        else:
            return name                      # unknown code: leave the name alone
    return f"{base}<{', '.join(args)}>"


_mangled_of = {}     # rocprofv3 display name -> mangled name (its own demangler garbles DF16b / DF16_ template arguments)


def load_symbols(db):
    for (t,) in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_info_kernel_symbol%'").fetchall():
        for kn, dn in db.execute(f"select kernel_name, display_name from {t}"):
            if kn and dn and kn.startswith("_Z"):
                _mangled_of[dn] = kn[:-3] if kn.endswith(".kd") else kn


def short(name):
    """kernel name as bench.py's profiler rows spell it: demangled, no return type, no argument list"""
    if name not in _demangled:
        src = _mangled_of.get(name, name)
        out = _demangle(src) if src.startswith("_Z") else src
        if out.startswith("_Z"):
            out = name
        out = re.sub(r"^void ", "", out)
        depth, cut = 0, len(out)
        for i, ch in enumerate(out):           # cut at the argument list's '(' (outside template brackets)
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        _demangled[name] = out[:cut].strip()
    return _demangled[name]


SETUP = ("__amd_rocclr_", "at::native::", "pack_weights_kernel")      # model set-up (load_state_dict copies, fills, weight packing): not the step


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    load_symbols(db)
    rows = db.execute("select name, duration from kernels").fetchall()
    agg = defaultdict(list)
    for n, d in rows:
        if any(short(n).startswith(p) for p in SETUP):
            continue
        agg[short(n)].append(d)
    tot = sum(sum(v) for v in agg.values())
    out = []
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append(dict(kernel=n, calls=len(v), total_ns=sum(v), avg_ns=round(sum(v) / len(v), 1), min_ns=min(v), max_ns=max(v),
                        percent=round(100.0 * sum(v) / tot, 2)))
    return out


def pmc_avg(db_path, counter):
    db = sqlite3.connect(db_path)
    load_symbols(db)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = defaultdict(list)
    for n, v in rows:
        agg[short(n)].append(v)
    return {n: (sum(v) / len(v), len(v)) for n, v in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outdir = os.environ.get("MFC_PROFILES_DIR") or os.path.join(root, "profiles")      # (on the GPU box: somewhere under gpurun_out/, which travels back)
    os.makedirs(outdir, exist_ok=True)
    for mode in ("serial", "lanes"):
        p = os.path.join(src, mode, f"{mode}_results.db")
        if not os.path.exists(p):
            continue
        st = kernel_stats(p)
        with open(os.path.join(outdir, f"{tag}_kernel_stats_{mode}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(st[0].keys()))
            w.writeheader(); w.writerows(st)
        print(mode, "top kernels:")
        for r in st[:8]:
            print(f"  {r['percent']:6.2f}%  n={r['calls']:5d}  avg {r['avg_ns'] / 1e3:8.1f} us  {r['kernel'][:90]}")
    pm = os.path.join(src, "pmc_mfma", "mfma_results.db")
    if os.path.exists(pm):
        # MFMA utilisation (north_star): SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes were busy, summed over all SIMDs;
        # GRBM_GUI_ACTIVE = busy clocks summed over the 8 XCDs.  util = MFMA busy / (kernel clocks x 1024 SIMDs).
        mb, sb, ga = pmc_avg(pm, "SQ_VALU_MFMA_BUSY_CYCLES"), pmc_avg(pm, "SQ_BUSY_CYCLES"), pmc_avg(pm, "GRBM_GUI_ACTIVE")
        out = {"note": "per-dispatch averages over one serial training step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE); "
                       "mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); mfma_busy_over_sq_busy = the raw counter ratio",
               "kernels": {}}
        for n, (v, calls) in mb.items():
            gui = ga.get(n, (0.0, 0))[0]
            sq = sb.get(n, (0.0, 0))[0]
            out["kernels"][n] = {"dispatches": calls, "mfma_busy_cycles": round(v), "sq_busy_cycles": round(sq), "grbm_gui_active": round(gui),
                                 "mfma_util": round(v / (gui / 8.0 * 1024.0), 4) if gui else None,
                                 "mfma_busy_over_sq_busy": round(v / sq, 4) if sq else None}
        with open(os.path.join(outdir, f"{tag}_pmc_mfma.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        for n in sorted(out["kernels"], key=lambda k: -out["kernels"][k]["mfma_busy_cycles"] * out["kernels"][k]["dispatches"])[:8]:
            k = out["kernels"][n]
            print(f"  mfma {k['dispatches']:5d} x util {k['mfma_util']}  busy/sq_busy {k['mfma_busy_over_sq_busy']}  {n[:90]}")
    pf, pw = os.path.join(src, "pmc_fetch", "fetch_results.db"), os.path.join(src, "pmc_write", "write_results.db")
    if os.path.exists(pf) and os.path.exists(pw):
        fe, wr = pmc_avg(pf, "FETCH_SIZE"), pmc_avg(pw, "WRITE_SIZE")
        out = {"note": "per-dispatch averages over one serial training step; FETCH_SIZE doubled (gfx950 wide-read correction), units of 1024 B",
               "kernels": {}}
        nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # training steps of the profiled command (tools/profile_round.sh: --warmup 1 --steps 1)
        step_bytes = 0.0
        for n in fe:
            f_kb, calls = fe[n]
            w_kb = wr.get(n, (0.0, 0))[0]
            out["kernels"][n] = {"dispatches": calls, "fetch_size_raw_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                                 "hbm_bytes_per_launch": round((2.0 * f_kb + w_kb) * 1024.0)}
            if not any(n.startswith(p) for p in SETUP):
                step_bytes += (2.0 * f_kb + w_kb) * 1024.0 * calls
        out["step"] = {"steps_in_run": nsteps, "hbm_bytes_per_step": round(step_bytes / nsteps),
                       "note": "counter bytes of every kernel of the run except model set-up (copies, fills, weight packing), divided by the steps of the run"}
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        try:
            from bench import code_hash                      # (what the table was taken on: bench.py marks it stale when the sources move on)
            out["code_hash"] = code_hash()
        except Exception as e:                                # noqa: BLE001
            out["code_hash"] = None
        with open(os.path.join(outdir, f"{tag}_pmc_traffic.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        for n in sorted(out["kernels"], key=lambda k: -out["kernels"][k]["hbm_bytes_per_launch"] * out["kernels"][k]["dispatches"])[:8]:
            k = out["kernels"][n]
            print(f"  pmc {k['dispatches']:5d} x {k['hbm_bytes_per_launch'] / 1e6:9.2f} MB  {n[:90]}")


if __name__ == "__main__":
    main()
