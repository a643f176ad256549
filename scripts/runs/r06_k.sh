python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1_split" 2>&1 | tail -3
for v in 1 2; do
HEAL_SPLIT_V=$v python scripts/split_gemm_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$v', [ (r['f32_mfma']['us'], r['bf16x6']['us'], r['bf16x9']['us']) for r in d['rows']])"
done
