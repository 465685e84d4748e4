( MLP_PUMP_DEBUG=1 MLP_TRANSPORT=pump MLP_NO_GRAPH=1 timeout 120 python tools/shard_test.py 2 3000 3000 12 100 sparse 2>&1 | grep -v "Warn\|^\[W" | tail -25 | cut -c1-250 ) 2>&1 | sed "s/^/pump-nograph: /"
( MLP_PUMP_DEBUG=1 MLP_TRANSPORT=pump timeout 120 python tools/shard_test.py 2 3000 3000 12 100 sparse 2>&1 | grep -v "Warn\|^\[W" | tail -25 | cut -c1-250 ) 2>&1 | sed "s/^/pump-graph: /"
