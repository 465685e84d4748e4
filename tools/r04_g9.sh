mkdir -p gpurun_out/r04
( timeout 900 python tools/transport_200k.py 20000 20000 4 0.4 --paths factor,oracle --json gpurun_out/r04/g9_transport_40k.json 2>&1 | grep -v "Warn\|^\[W" | tail -30 ) 2>&1 | sed "s/^/40k: /"
( timeout 1500 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor --json gpurun_out/r04/g9_transport_200k.json 2>&1 | grep -v "Warn\|^\[W" | tail -30 ) 2>&1 | sed "s/^/200k: /"
