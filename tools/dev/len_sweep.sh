# packed-route segment time by read length: default routing, the wave kernel, (SoA) the LDS-tiled kernel
for spec in "1000000 40" "1000000 75" "1000000 150" "500000 300" "250000 600"; do
  set -- $spec
  echo "== $1 x $2 bp"
  python3 tools/wave_time.py $1 $2 10 2>&1 | tail -1
  VGAN_HC_KERNEL=wave python3 tools/wave_time.py $1 $2 10 2>&1 | tail -1
done
