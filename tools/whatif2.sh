#!/bin/bash
# what-if timings of the W32 step under lanes: step time with one record kind skipped (results are wrong, timing only)
cd /root/repo
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-prof"
run() { echo -n "$1: "; env $2 $B $3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms', d['value'], 'frames/s')"; }
run "baseline" "X=1"
run "no wgrad (kinds 2,17)" "MFC_SKIP_KINDS=$(( (1<<2) | (1<<17) | (1<<14) ))"
run "no bnbwd reduce (5)" "MFC_SKIP_KINDS=$(( 1<<5 ))"
run "no bnbwd apply (7)" "MFC_SKIP_KINDS=$(( 1<<7 ))"
run "no bnbwd reduce+apply+fin" "MFC_SKIP_KINDS=$(( (1<<5) | (1<<6) | (1<<7) ))"
run "no bnfin fwd (3)" "MFC_SKIP_KINDS=$(( 1<<3 ))"
run "no combine+maskadd (4,8)" "MFC_SKIP_KINDS=$(( (1<<4) | (1<<8) ))"
run "no conv (1)" "MFC_SKIP_KINDS=$(( 1<<1 ))"
run "serial" "X=1" "--serial"
run "batch 1" "X=1" "--batch 1"
run "batch 1 serial" "X=1" "--batch 1 --serial"
run "fwd only" "X=1" "--fwd-only"
