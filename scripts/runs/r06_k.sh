for t in 1 2 3; do
HEAL_SPLIT_TILE=$t python scripts/split_gemm_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $t', [ (r['f32_mfma']['us'], r['bf16x6']['us'], r['bf16x9']['us']) for r in d['rows']])"
done
HEAL_SPLIT_TILE=2 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1_split" 2>&1 | tail -2
