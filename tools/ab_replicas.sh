#!/bin/bash
# A/B of MFC_STAT_REPLICAS on one box: expects mfcnet-tracker_amd/csrc/build/variants/libR<n>.so (build each with the header constant set to n).
# The shipped library and _lib.py are swapped for the duration of the run only: both are saved first and restored on ANY exit.
cd /root/repo
LIB=mfcnet-tracker_amd/mfcnet_amd/libmfcnet_hip.so
PY=mfcnet-tracker_amd/mfcnet_amd/_lib.py
SAVE=$(mktemp -d)
cp -p $LIB $SAVE/lib.so && cp -p $PY $SAVE/_lib.py || { echo "cannot save $LIB / $PY"; exit 1; }
restore() { cp -p $SAVE/lib.so $LIB; cp -p $SAVE/_lib.py $PY; rm -rf $SAVE; }
trap restore EXIT
one() { python bench.py "$@" --no-cpu-baseline --no-prof --no-fp16-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
for R in 32 16 8; do
  cp mfcnet-tracker_amd/csrc/build/variants/libR$R.so $LIB
  sed -i "s/^STAT_REPLICAS = [0-9]*/STAT_REPLICAS = $R/" $PY
  echo "R=$R  W32 B8: $(one --steps 20 --warmup 5)  b=1: $(one --batch 1 --steps 30 --warmup 5)  W48: $(one --width 48 --steps 10 --warmup 3)"
done
done
