cd tools/micro
for cfg in "8 1 0 0 16" "8 1 16 16 16" "8 1 16 0 16" "8 1 0 16 16" "8 1 17 17 17" "4 2 16 16 16" "2 4 16 16 16" "1 8 16 16 16" "4 1 16 16 16" "4 1 0 0 16" "1 1 0 0 16" "1 1 16 16 16" "4 2 0 0 16" "4 2 16 16 2"; do
  timeout -k 5 60 ./xcdsplit 3000 $cfg 0 | tail -1 || exit 1
done
for cfg in "8 1 0 0 16" "8 1 16 16 16" "4 2 16 16 16" "2 4 16 16 16" "1 8 16 16 16"; do
  timeout -k 5 60 ./xcdsplit 3000 $cfg 4 | tail -1 || exit 1
done
