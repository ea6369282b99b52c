#!/bin/bash
# lane -> stream maps (mfc_set_flag(12): digits = streams of the module branches 1..4; 0 = library default) on the full step
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
for w in 32 48; do
  run "w$w default (1,2,3*,1)" "X=1" "--width $w"
  for m in 1221 1231 1232 1233 1234 1212; do
    run "w$w map $m" "MFC_LANE_STREAMS=$m" "--width $w"
  done
done
MFC_DEBUG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep "mfc lanes"
