#!/bin/bash
# Everything a round's profiles/ are condensed from, in one gpurun call (run from the repo root on the GPU box):
#   rocprofv3 kernel stats + PMC passes for W32 and W48, the default bench lines, per-record profiles, what-if runs, the training soak
set -e
O=gpurun_out/round${WIDTHS:+_w$WIDTHS}; rm -rf $O; mkdir -p $O
TAG=${1:-r03_e}
# (the rocprofv3 databases are ~60 MB per width and gpurun returns at most 64 MB: condense them here, keep only the small tables)
for w in ${WIDTHS:-32 48}; do
  bash tools/profile_round.sh $w > $O/profile_round_w$w.log 2>&1
  MFC_PROFILES_DIR=$O/profiles python tools/summarize_profile.py gpurun_out/prof_round ${TAG}_w$w 2 > $O/summarize_w$w.log 2>&1
  cp gpurun_out/prof_round/bench_*.log $O/ 2>/dev/null || true
  for f in $O/bench_serial.log $O/bench_lanes.log $O/bench_pmc_fetch.log $O/bench_pmc_write.log $O/bench_pmc_mfma.log; do [ -f $f ] && mv $f ${f%.log}_w$w.log; done
  rm -rf gpurun_out/prof_round
done
cp $O/profiles/${TAG}_w*_pmc_traffic.json profiles/        # (bench.py looks the counter bytes up in the newest committed PMC table)
[ -n "$ONLY_PROFILE" ] && exit 0
python bench.py --steps 20 --warmup 5 > $O/bench_w32.json 2> $O/bench_w32.err
python bench.py --width 48 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_w48.json 2> $O/bench_w48.err
python bench.py --dtype fp16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_w32_fp16.json 2> $O/bench_w32_fp16.err
python tools/profile_step.py 32 4 > $O/step_by_record_w32.txt 2>&1
python tools/profile_step.py 48 2 > $O/step_by_record_w48.txt 2>&1
bash tools/whatif.sh > $O/whatif_w32.txt 2>&1
for dt in fp32 fp16 bf16; do python tools/train_soak.py $dt 150 32 > $O/soak_$dt.txt 2>&1; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -c 400 $O/bench_w32.json
