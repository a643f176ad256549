# split-bf16 prototype: final numbers -> gpurun_out/r06/r06_split_gemm.json
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1_split" 2>&1 | tail -2
python scripts/split_gemm_bench.py > gpurun_out/r06/split_gemm_bench.json 2>/dev/null
for ar in bf16x6 bf16x9; do
HEAL_ARITH=$ar python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "config4_full or config5_full" 2>&1 | tail -1
HEAL_ARITH=$ar python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_$ar.json 2>/dev/null
HEAL_ARITH=$ar python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload scene8_second_v2xvit > gpurun_out/r06/bench8_$ar.json 2>/dev/null
done
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/bench_f32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload scene8_second_v2xvit > gpurun_out/r06/bench8_f32.json 2>/dev/null
python - <<'PY'
import json
out=json.load(open('gpurun_out/r06/split_gemm_bench.json'))
par=[json.loads(l) for l in open('gpurun_out/parity_report.jsonl') if 'bf16x' in l or 'conv1x1_split_error' in l]
out['error_criterion_rows']=[p for p in par if p['test']=='conv1x1_split_error'][-12:]
out['end_to_end_parity']=[p for p in par if p['test'].startswith('config')][-4:]
e2e={}
for tag in ('f32','bf16x6','bf16x9'):
    for wl,pre in (('scene5','bench_'),('scene8_second_v2xvit','bench8_')):
        try:
            d=json.load(open(f'gpurun_out/r06/{pre}{tag}.json'))
            e2e[f'{wl}_{tag}']={'scenes_per_s':d['value'],'ms_per_step':d['ms_per_step'],'dtype':d['dtype'][:40]}
        except Exception as e: e2e[f'{wl}_{tag}']=str(e)
out['end_to_end_rate']=e2e
json.dump(out,open('gpurun_out/r06/r06_split_gemm.json','w'),indent=1)
print(json.dumps(e2e,indent=0)); print(out['end_to_end_parity'])
PY
