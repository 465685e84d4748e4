for fb in 8192 4096 3072 2048 1536 1024; do echo "MLP_FOLD_BLOCKS=$fb"; MLP_FOLD_BLOCKS=$fb timeout 200 python tools/reinvert_timing.py late 2 2>&1 | grep "late load 1" | cut -c1-120; done
timeout 400 python tools/ab_env.py late "MLP_FOLD_BLOCKS=8192" "MLP_FOLD_BLOCKS=4096" "MLP_FOLD_BLOCKS=2048" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
