#!/bin/bash
# sample clocks / power while the multi-stream and the serial bench run
cd /root/repo
for mode in "" "--serial"; do
  python bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-prof $mode > /tmp/b.json 2>/dev/null &
  BP=$!
  sleep 10
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|GPU use|fclk" | tr '\n' ';' ; echo
    sleep 1
  done
  wait $BP
  python -c "import json; d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0]); print('mode [$mode]', d['ms_per_step'], 'ms')"
done
