#!/bin/bash
# round 6: config 5 (SECOND + V2X-ViT, 8 agents) with the pair-tile thin layers, A/B against the neighbour-table path
mkdir -p gpurun_out/r06
HEAL_SP_TILES=0 timeout 600 python bench.py --workload scene8_second_v2xvit --no-cpu-baseline > gpurun_out/r06/bench_c5_table.json 2> gpurun_out/r06/bench_c5_table.err
timeout 600 python bench.py --workload scene8_second_v2xvit --no-cpu-baseline > gpurun_out/r06/bench_c5_tiles.json 2> gpurun_out/r06/bench_c5_tiles.err
python - <<'PY'
import json
for tag in ("table", "tiles"):
    try:
        d = json.loads(open(f"gpurun_out/r06/bench_c5_{tag}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(tag, "unreadable", e, open(f"gpurun_out/r06/bench_c5_{tag}.err").read()[-1500:]); continue
    k3 = [k for k in d["roofline_other"] if k["kernel"].startswith("K3")][0]
    print(tag, d["value"], d["ms_per_step"], "K3 step_ms", k3["step_ms"], "rulebook", k3["rulebook_ms"], "frac", k3["frac"], "with_rb", k3["frac_with_rulebook"])
    for L in k3["layers"][:3]:
        print("   ", L["cin"], L["cout"], L["us"], L["frac_hbm"])
PY
