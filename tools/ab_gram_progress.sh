#!/bin/bash
# Gram mode: objective gained per pivot from the k = 10 000 basis of config 4 — the streaming pass, the
# Gram path, the Gram path with the weights clamped, and the shadow diagnostic (Gram path + streaming pass in the same
# pivot: max |v_gram - v_stream|; mode 2 continues with the streamed v, i.e. M is maintained but not used for pricing)
echo "== stream (MLP_GRAM=0)"; MLP_GRAM=0 python tools/gram_progress.py mid 20000 2>&1 | tail -5
echo "== gram"; MLP_GRAM=1 MLP_GRAM_PROBE=1 python tools/gram_progress.py mid 20000 2>&1 | tail -6
echo "== gram + shadow 1 (measure)"; MLP_GRAM=1 MLP_GRAM_SHADOW=1 python tools/gram_progress.py mid 20000 2>&1 | tail -12
echo "== gram + shadow 2 (streamed v drives the weights)"; MLP_GRAM=1 MLP_GRAM_SHADOW=2 python tools/gram_progress.py mid 20000 2>&1 | tail -12
