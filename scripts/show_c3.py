import json
d = json.load(open("gpurun_out/r02_conv3x3_bench.json"))
for k, r in d.items():
    ds = " ".join(f"th{t}:{r[f'direct_th{t}_us']}" for t in (16, 8, 4) if f"direct_th{t}_us" in r)
    print(f"{k:<38} GF {r['GF']:>6} miopen {r['miopen+bias_act_us']:>7} | direct {ds:<34} | wino w4 {r.get('winograd_us', '-'):>6} w8 {r.get('winograd_w8_us', '-'):>6} | default {r.get('default_us')} ({r.get('default_algo')})")
