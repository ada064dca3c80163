# decoder 3x3 layers at full resolution: fwd and data gradient
for s in "304 128 0" "240 64 0" "144 96 0" "128 32 0" "128 304 1" "64 240 1" "96 144 1" "32 128 1"; do
  set -- $s
  CIN=$1 COUT=$2 python tools/bench_conv.py $3 20 | tail -1 | sed "s/^/mode $3: /"
done
