one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --no-fp16-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['step_ms']['median'])"; }
echo "base $(one)"
for b in 96 160 192 256; do echo "wgrad_blocks=$b $(MFC_WGRAD_BLOCKS=$b one)"; done
echo "ring_wgs=1 $(MFC_RING_WGS=1 one)"
echo "bnred=512 $(MFC_BNRED_BLOCKS=512 one)"
echo "bnred=2048 $(MFC_BNRED_BLOCKS=2048 one)"
echo "base $(one)"
