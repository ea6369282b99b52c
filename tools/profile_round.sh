#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the serial (one-stream) bench and of the default (lanes) bench
#   2. three separate PMC passes (FETCH_SIZE; WRITE_SIZE; MFMA busy cycles) of the serial bench  -- never combined with other trace domains
# Outputs land in gpurun_out/prof_round/ ; tools/summarize_profile.py condenses them into profiles/.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
W=${1:-32}
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/serial -o serial -- python3 /root/repo/bench.py --serial --width $W --steps 3 --warmup 2 --no-cpu-baseline --no-fp16-line > $OUT/bench_serial.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/lanes -o lanes -- python3 /root/repo/bench.py --width $W --steps 3 --warmup 2 --no-cpu-baseline --no-fp16-line > $OUT/bench_lanes.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python3 /root/repo/bench.py --serial --width $W --steps 1 --warmup 1 --no-cpu-baseline --no-fp16-line --no-prof > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python3 /root/repo/bench.py --serial --width $W --steps 1 --warmup 1 --no-cpu-baseline --no-fp16-line --no-prof > $OUT/bench_pmc_write.log 2>&1
# MFMA utilisation of the top kernels (north_star: "rocprof ... MFMA utilisation"): its own PMC pass, counters only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o mfma -- python3 /root/repo/bench.py --serial --width $W --steps 1 --warmup 1 --no-cpu-baseline --no-fp16-line --no-prof > $OUT/bench_pmc_mfma.log 2>&1
find $OUT -name "*.db" | head -20
