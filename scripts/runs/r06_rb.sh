#!/bin/bash
# round 6: the rulebook kernels after wave-aggregated bitmap marks: tests (bit-identical rulebooks) + per-kernel averages in the config-5 step
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "rulebook or rank or sparse or second_encoder or pair_tiles" 2>&1 | tail -3
bash scripts/k3_rulebook_prof.sh 2>&1 | grep -v "^$" | tail -30
