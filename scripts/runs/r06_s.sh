#!/bin/bash
# round 6: the heterogeneous scene sharded over two ranks with a camera-only rank (camera-crop walk inside ShardedCollab.local)
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "heterogeneous" 2>&1 | tail -15
