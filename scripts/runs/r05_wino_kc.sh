mkdir -p gpurun_out/r05
for kc in 8 16 8 16; do echo "== HEAL_WG_KC=$kc"; HEAL_WG_KC=$kc timeout 120 python scripts/wino_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/wino_kc16.txt
