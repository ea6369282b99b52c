#!/bin/bash
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s', 'loss', d['config']['final_loss'])"; }
for w in 32 48; do
  run "w$w fused" "MFC_FUSE_BNRED=1" "--width $w"
  run "w$w unfused" "MFC_FUSE_BNRED=0" "--width $w"
done
run "w32 fused serial" "MFC_FUSE_BNRED=1" "--width 32 --serial"
run "w32 unfused serial" "MFC_FUSE_BNRED=0" "--width 32 --serial"
run "w32 fused b1" "MFC_FUSE_BNRED=1" "--width 32 --batch 1"
