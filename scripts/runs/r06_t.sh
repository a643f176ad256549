#!/bin/bash
# round 6: the K3 pair-tile kernel (k_sp_tiles) against k_sp_conv2 on the first layers of config 5; ANATOMY="0 7 8" (HEAL_SP_TILES_DBG) needs
# a HEAL_BUILD_EXPERIMENTAL=1 library
mkdir -p gpurun_out/r06
if [ -n "$ANATOMY" ]; then
  for d in ${ANATOMY}; do
    echo "== HEAL_SP_TILES_DBG=$d"
    HEAL_SP_TILES_DBG=$d timeout 300 python scripts/k3_bench.py --layers 3 --iters 30 --modes ${MODES:-tiles} --no-check --brief --graph 2>&1 | grep "us$"
  done
else
  timeout 600 python scripts/k3_bench.py --layers ${LAYERS:-3} --iters 30 --modes ${MODES:-v2,tiles} --graph --json gpurun_out/r06/k3_thin.json 2>&1 | tail -${TAIL:-6}
fi
