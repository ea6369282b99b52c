#!/bin/bash
# the other BASELINE.json points with the same bench.py (1 x MI355X, bf16): configs[1], eval forward, per-GPU shares of configs[3] / configs[4]
cd /root/repo
O=gpurun_out/bench_configs; mkdir -p $O
python bench.py --single --fwd-only --steps 30 --warmup 5 --no-cpu-baseline --no-fp16-line > $O/cfg1_single_fwd_w32.json 2>/dev/null
python bench.py --single --fwd-only --width 48 --steps 30 --warmup 5 --no-cpu-baseline --no-fp16-line > $O/cfg1_single_fwd_w48.json 2>/dev/null
python bench.py --fwd-only --steps 30 --warmup 5 --no-cpu-baseline --no-fp16-line > $O/mfcnet_fwd_w32.json 2>/dev/null
python bench.py --batch 4 --depth --optflow --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-line > $O/cfg3_share_w32.json 2>/dev/null
python bench.py --batch 4 --depth --optflow --basic --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-line > $O/cfg3_share_basic_w32.json 2>/dev/null
python bench.py --width 48 --frames 5 --height 720 --width-px 960 --steps 8 --warmup 3 --no-cpu-baseline --no-fp16-line > $O/cfg4_share_w48_t5_720x960.json 2>/dev/null
python bench.py --width 48 --frames 5 --height 720 --width-px 960 --dtype fp16 --steps 8 --warmup 3 --no-cpu-baseline --no-prof > $O/cfg4_share_w48_t5_720x960_fp16.json 2>/dev/null
python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof --no-fp16-line > $O/b1_w32.json 2>/dev/null
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print(sys.argv[1].split('/')[-1], d['value'], d['unit'], d['ms_per_step'], 'ms |', (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))
PY
done
