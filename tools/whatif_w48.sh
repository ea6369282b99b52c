# what-if timing for HRNet-W48: skip record kinds (bit k = kind k: 1 conv, 2 wgrad, 3 bnfin, 4 combine, 5 reduce, 6 bnbwd_fin, 7 apply, 8 mask_add)
for m in 0 4 8 64 16 32 128 256 2; do
  r=$(MFC_SKIP_KINDS=$m timeout -k 10 300 python bench.py --width 48 --no-cpu-baseline --no-prof --no-fp16-line --steps 10 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/')
  echo "skip_mask=$m -> $r ms/step"
done
