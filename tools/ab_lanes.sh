for w in late mid; do
  for l in 64 16 4; do
    MLP_LANES=$l python tools/window_profile.py $w 512 2>&1 | grep pivots/s | sed "s/^/lanes $l: /"
  done
done
for l in 64 16; do MLP_LANES=$l python tools/window_profile.py early 2000 200 2>&1 | grep pivots/s | sed "s/^/lanes $l: /"; done
