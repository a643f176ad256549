#!/bin/bash
# round 6: K3 thin-layer kernel (k_sp_thin) against k_sp_conv2 on the three CIN <= 16 layers of config 5
mkdir -p gpurun_out/r06
if [ -n "$ANATOMY" ]; then
  for d in 0 1 2 4 7 8 16; do
    echo "== HEAL_SP_THIN_DBG=$d"
    HEAL_SP_THIN_DBG=$d timeout 300 python scripts/k3_bench.py --layers 3 --iters 30 --modes thin --no-check --brief 2>&1 | grep "us$"
  done
else
  timeout 600 python scripts/k3_bench.py --layers 3 --iters 30 --modes ${MODES:-v2,thin,tiles:d2,tiles:d4,tiles:d8} --json gpurun_out/r06/k3_thin.json 2>&1 | tail -12
fi
