for map in 1212 1222 1221 1122 1232 1233 1223 1231 1123; do
  r=$(MFC_OWN_MAIN=0 MFC_LANE_STREAMS=$map timeout -k 10 300 python bench.py --no-cpu-baseline --no-prof --steps 10 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
  echo "map=$map -> $r"
done
