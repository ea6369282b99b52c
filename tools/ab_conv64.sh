#!/bin/bash
cd /root/repo
python tools/bench_conv.py bf16 fwd 2>/dev/null | head -12
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s', d['config']['final_loss'])"; }
run "w32" "X=1" "--width 32"
run "w32 serial" "X=1" "--width 32 --serial"
run "w48" "X=1" "--width 48"
