#!/bin/bash
# small knobs of the serial chain around the streaming pass (late / mid windows)
for w in late mid; do
  for c in 24 30 36 48; do
    MLP_PB_CHUNKS=$c python tools/window_profile.py $w 512 64 2>&1 | grep pivots/s | sed "s/^/pb_chunks $c: /"
  done
done
timeout 300 python -m pytest tests/test_lowrank.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -3
