for cfg in "MLP_FACTOR_GJ_LAUNCHES=1" "MLP_FACTOR_SKIP=0" "MLP_X=1"; do
( env $cfg timeout 600 python tools/transport_200k.py 30000 50000 4 0 --family mixed --paths factor --chunk 20000 2>&1 | grep -v Warn | grep "^factor" | cut -c1-330 ) 2>&1 | sed "s/^/$cfg: /"
done
