#!/bin/bash
echo "== stream (MLP_GRAM=0)"; MLP_GRAM=0 python tools/gram_progress.py mid 20000 2>&1 | tail -5
echo "== gram default"; MLP_GRAM_PROBE=1 python tools/gram_progress.py mid 20000 2>&1 | tail -6
echo "== gram, safeguard always"; MLP_GRAM_SAFE=-1 python tools/gram_progress.py mid 20000 2>&1 | tail -5
