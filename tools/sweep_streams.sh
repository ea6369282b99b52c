# sweep the lane -> stream map / detached stream / wgrad workgroups on the full step (usage: bash tools/sweep_streams.sh [width])
W=${1:-32}
for map in 1221 1212 1222 1231; do for blk in 256 512; do
  r=$(MFC_LANE_STREAMS=$map MFC_WGRAD_BLOCKS=$blk timeout -k 10 300 python bench.py --width $W --no-cpu-baseline --no-prof --steps 10 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
  echo "width=$W map=$map wgrad_blocks=$blk -> $r"
done; done
