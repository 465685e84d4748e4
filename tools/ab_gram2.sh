#!/bin/bash
python tools/gram_check.py 2>&1 | grep -v "gram=0" | tail -8
for w in late mid; do
  MLP_GRAM=1 python tools/window_profile.py $w 512 64 2>&1 | grep pivots/s | sed "s/^/gram 1: /"
done
python tools/gram_drift.py late 6 512 2>&1 | tail -6
