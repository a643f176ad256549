# anatomy of k_pillar_stem2 (HEAL_PS_DBG bits skip parts: timing only)
for d in 0 1 2 3 4 7; do
HEAL_PS_DBG=$d python scripts/pillar_stem_bench.py 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $d', r['kernel_own_us'].get('pillar_stem_block'), r['kernel_own_us'].get('pfn_pillars'), round(r['pillar_chain_us'],1))"
done
