#!/bin/bash
# W48 knob sweep on the final round-4 structure (one gpurun call): bash tools/sweep_w48_knobs.sh > gpurun_out/r04_w48_knobs.txt
B="python bench.py --width 48 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-prof"
run() { echo -n "$1: "; env $2 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
run "default" "X=1"
run "default again" "X=1"
for v in 96 192 256; do run "wgrad blocks $v" "MFC_WGRAD_BLOCKS=$v"; done
for v in 0 100 110; do run "nw8 weight $v" "MFC_CONV_NW8=$v"; done
run "write-through from 4 MB" "MFC_WT_MIN_MB=4"
run "write-through from 48 MB" "MFC_WT_MIN_MB=48"
run "bnred blocks 512" "MFC_BNRED_BLOCKS=512"
run "applyfin blocks 512" "MFC_APPLYFIN_BLOCKS=512"
run "wgrad dma48 off" "MFC_WGRAD_DMA48=0"
run "lane4 fwd off" "MFC_LANE4_FWD=0"
run "default third" "X=1"
