#!/bin/bash
# round 6: the pair-tile path -- kernel tests, the SECOND encoder golden, the config-5 model tests
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "pair_tiles or sparse or second_encoder" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_train.py -q -x -k "second or sparse or SECOND or config5 or baseline" 2>&1 | tail -8
