#!/bin/bash
# only the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the serial W32 bench and the condensed traffic table -> gpurun_out/pmc_only/ (a full set: tools/round_measure.sh)
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_round; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/pmc_only
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python3 /root/repo/bench.py --serial --width 32 --steps 1 --warmup 1 --no-cpu-baseline --no-fp16-line --no-prof > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python3 /root/repo/bench.py --serial --width 32 --steps 1 --warmup 1 --no-cpu-baseline --no-fp16-line --no-prof > $OUT/bench_pmc_write.log 2>&1
cd $R
MFC_PROFILES_DIR=$R/gpurun_out/pmc_only python tools/summarize_profile.py gpurun_out/prof_round r04_l_w32 2 > gpurun_out/pmc_only/summarize.log 2>&1
rm -rf $OUT
ls gpurun_out/pmc_only
