mkdir -p gpurun_out/r04
( timeout 2400 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor,dense,oracle --dense-pivots 60000 --json gpurun_out/r04/transport_200k_evidence.json 2>&1 | grep -v "Warn\|^\[W" | tail -40 | cut -c1-300 )
