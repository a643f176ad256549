for d in 0 1 2 4 8 15; do
HEAL_SPLIT_DBG=$d python scripts/split_gemm_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d', [ (r['bf16x6']['us'], r['bf16x9']['us']) for r in d['rows'][:4]])"
done
