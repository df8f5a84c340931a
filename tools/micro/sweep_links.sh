# cost of each link of the whole-loop kernel's per-step exchange, by leaving links out (tools/micro/xcdsplit.hip)
cd tools/micro
for shots in 8 4; do
for cfg in "0 1" "1 1" "2 1" "4 1" "6 1" "3 1" "7 1" "8 1" "15 1" "0 2" "0 4"; do
  set -- $cfg
  timeout -k 5 60 ./xcdsplit 3000 $shots 1 0 0 16 0 $1 $2 | tail -1 || exit 1
done
done
for cfg in "0 1" "0 2"; do set -- $cfg; timeout -k 5 60 ./xcdsplit 3000 4 1 0 0 16 4 $1 $2 | tail -1 || exit 1; done
