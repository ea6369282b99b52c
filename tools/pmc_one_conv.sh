#!/bin/bash
# HBM traffic / MFMA-busy counters of ONE convolution shape (separate rocprofv3 --pmc passes; run through gpurun from the repo root):
#   tools/pmc_one_conv.sh N Cin Cout k s H W [fused]
# prints per-kernel averages: FETCH_SIZE (x2: MI355X_MICROARCH.md "HBM"), WRITE_SIZE, MFMA busy share
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace -d $OUT/$T -o p -- python3 $ROOT/tools/one_conv.py "$@" > $OUT/$T.log 2>&1
done
python3 - <<PY
import sqlite3, glob, collections
for db in sorted(glob.glob("$OUT/*/p_results.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    acc = collections.defaultdict(list)
    for k, n, v in rows:
        if "conv" in k: acc[(k, n)].append(v)
    for (k, n), vs in sorted(acc.items()):
        print(f"{n:28s} avg {sum(vs)/len(vs):14.1f}  n={len(vs):3d}  {k[:80]}")
PY
