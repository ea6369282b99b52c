#!/bin/bash
# smaller convolution footprints (pixel tile MT = 2, LDS budget) on the multi-stream and the serial step
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
run "default" "X=1" ""
run "MT=2" "MFC_CONV_MT=2" ""
run "MT=2 serial" "MFC_CONV_MT=2" "--serial"
run "LDS 52K" "MFC_CONV_LDS_KB=52" ""
run "LDS 52K serial" "MFC_CONV_LDS_KB=52" "--serial"
run "MT=2 LDS 40K" "MFC_CONV_MT=2 MFC_CONV_LDS_KB=40" ""
run "no NW8" "MFC_CONV_NW8=0" ""
run "no NW8 serial" "MFC_CONV_NW8=0" "--serial"
run "MT=2 no NW8" "MFC_CONV_MT=2 MFC_CONV_NW8=0" ""
