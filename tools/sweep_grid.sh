#!/bin/bash
# persistent conv workgroups per launch (mfc_set_flag(4)) x weight-gradient workgroups (flag 11) on the full step
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
for g in 256 320 384 448 512 640 768; do
  for b in 128 256; do
    run "w32 conv_grid=$g wgrad_blocks=$b" "MFC_CONV_GRID=$g MFC_WGRAD_BLOCKS=$b" "--width 32"
  done
done
run "w32 conv_grid=256 serial" "MFC_CONV_GRID=256" "--width 32 --serial"
run "w32 conv_grid=512 serial" "MFC_CONV_GRID=512" "--width 32 --serial"
