# round 6, call H: split-bf16 pointwise convolution -- error criterion, kernel bench, end-to-end parity + rate under HEAL_ARITH
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1_split" 2>&1 | tail -8
python scripts/split_gemm_bench.py > gpurun_out/r06/split_gemm_bench.json 2> gpurun_out/r06/split_gemm_bench.err; tail -2 gpurun_out/r06/split_gemm_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06/split_gemm_bench.json'))
for r in d['rows']: print(r['shape'], r['f32_mfma']['us'], r['bf16x6']['us'], r['bf16x9']['us'], '%.1e %.1e %.1e' % (r['f32_mfma']['max_rel_err_vs_fp64'], r['bf16x6']['max_rel_err_vs_fp64'], r['bf16x9']['max_rel_err_vs_fp64']), r['bf16x6']['frac_of_bf16_peak'])"
for ar in bf16x6 bf16x9; do
HEAL_ARITH=$ar python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "config4_full" 2>&1 | tail -2
HEAL_ARITH=$ar python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ar', d['value'], d['ms_per_step'], d['serial']['ms_per_step'])"
done
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', d['value'], d['ms_per_step'], d['serial']['ms_per_step'])"
grep "arith\|split\|bf16" gpurun_out/parity_report.jsonl | tail -20
