for d in 0 16 32 48 1; do
HEAL_PS_DBG=$d python scripts/pillar_stem_bench.py 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d', r['kernel_own_us'].get('pillar_stem_block'), round(r['pillar_chain_us'],1))"
done
