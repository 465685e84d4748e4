#!/bin/bash
# tile-shape experiment for the fused W pass (rebuilds on the GPU box)
for TR in 8 16 32; do
  sed -i "s/^constexpr int FW_TR = [0-9]*;/constexpr int FW_TR = $TR;/" minilp_amd/csrc/kernels.h
  python minilp_amd/build.py --force > /dev/null 2>&1
  echo "== FW_TR=$TR"
  python tools/gpu_perf.py 100000 100000 100 2500 8 prof 2>&1 | grep chunk | sed -n '1p;4p;8p' | cut -c1-200
done
sed -i "s/^constexpr int FW_TR = [0-9]*;/constexpr int FW_TR = 16;/" minilp_amd/csrc/kernels.h
