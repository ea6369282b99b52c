#!/bin/bash
# step time vs the weight-gradient workgroup target (mfc_set_flag(11)) and kernel choice (flag 29), W32 and W48
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
for w in 32 48; do
  for b in 96 128 192 256 384; do
    run "w$w dma blocks=$b" "MFC_WGRAD_BLOCKS=$b" "--width $w"
  done
  run "w$w old blocks=128" "MFC_WGRAD_BLOCKS=128 MFC_WGRAD_DMA=0" "--width $w"
done
