#!/bin/bash
# balanced strips of the streaming pass (one equal tile per co-resident block) against the fixed 128-row strips
# usage: ab_balanced.sh [block counts...]   (0 = fixed strips, 1 = occupancy x CUs)
VALS=${@:-"1 0 768"}
for w in late mid; do
  for b in $VALS; do
    MLP_STREAM_BALANCED=$b python tools/window_profile.py $w 512 64 2>&1 | grep pivots/s | sed "s/^/balanced $b: /"
  done
done
