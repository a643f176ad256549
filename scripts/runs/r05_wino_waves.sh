for wv in 8 4; do echo "== HEAL_WG_WAVES=$wv"; HEAL_WG_WAVES=$wv timeout 120 python scripts/wino_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; done
