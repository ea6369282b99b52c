#!/bin/bash
# Re-check of the step-level switches on whatever the CURRENT structure is (one gpurun call, ~8 GPU-minutes):
#   bash tools/sweep_final_knobs.sh [width] > gpurun_out/knobs.txt
# Round 4 learnt that a switch's optimum is a property of the schedule around it (switch 55 flipped sign after the fuse stage got one join per
# module and the 64-channel data gradients moved to the ring launch): run this at the END of a round, after the last structural change.
W=${1:-32}
B="python bench.py --width $W --steps 30 --warmup 8 --no-cpu-baseline --no-fp32-parity --no-prof"
run() { echo -n "$1: "; env $2 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
run "default" "X=1"
for v in 0 -1 110; do run "switch 55 (8-wave weight of fused data gradients) = $v" "MFC_CONV_NW8_FUSED=$v"; done
run "default" "X=1"
for v in 0 80 110; do run "switch 19 (8-wave weight, all launches) = $v" "MFC_CONV_NW8=$v"; done
for v in 128 192; do run "switch 11 (weight-gradient workgroups) = $v" "MFC_WGRAD_BLOCKS=$v"; done
run "default" "X=1"
for v in 50 300 100000; do run "switch 57 (ring launch for 64-channel data gradients from k pixels) = $v" "MFC_RING_C64_UNFUSED_KPX=$v"; done
for v in 4 48; do run "switch 54 (write-through stores from MB) = $v" "MFC_WT_MIN_MB=$v"; done
run "switch 56 (forward lane 4 on the detached queue) = 0" "MFC_LANE4_FWD=0"
run "switch 33 (ring workgroups per CU) = 3" "MFC_RING_WGS=3"
run "one join per module off" "MFC_ONE_JOIN=0"
run "merged stride-2 data gradients off" "MFC_MERGE_S2=0"
run "default" "X=1"
