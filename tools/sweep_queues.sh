#!/bin/bash
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>&1 | python -c "
import sys,json
ls=sys.stdin.read().splitlines()
p=[l for l in ls if 'mfc lanes' in l]
d=json.loads([l for l in ls if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s', p[-1][-60:] if p else '')"; }
for q in 4 5 6 8; do
  run "w32 GPU_MAX_HW_QUEUES=$q" "GPU_MAX_HW_QUEUES=$q MFC_DEBUG=1" "--width 32"
done
run "w32 q6 map 1231" "GPU_MAX_HW_QUEUES=6 MFC_LANE_STREAMS=1231" "--width 32"
run "w48 GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=4 MFC_DEBUG=1" "--width 48"
run "w48 GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=6 MFC_DEBUG=1" "--width 48"
