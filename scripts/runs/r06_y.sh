#!/bin/bash
# round 6: the pair-tile tests on the shipped library (128-site slots skipped) and on a HEAL_BUILD_EXPERIMENTAL=1 library built on the box
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pair_tiles or sparse or second_encoder or split_k" 2>&1 | tail -3
HEAL_BUILD_EXPERIMENTAL=1 python -c "from heal_amd import build; build.build()" > /tmp/exp_build.log 2>&1; tail -2 /tmp/exp_build.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pair_tiles" 2>&1 | tail -3
HEAL_SP_SLOT_SITES=128 MODES=tiles LAYERS=3 TAIL=2 bash scripts/runs/r06_t.sh | cut -c1-300
