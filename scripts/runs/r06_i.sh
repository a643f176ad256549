mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv1x1_split" 2>&1 | tail -3
python scripts/split_gemm_bench.py > gpurun_out/r06/split_gemm_bench.json 2> gpurun_out/r06/split_gemm_bench.err; tail -2 gpurun_out/r06/split_gemm_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06/split_gemm_bench.json'))
for r in d['rows']: print(r['shape'], r['f32_mfma']['us'], r['bf16x6']['us'], r['bf16x9']['us'], '%.1e %.1e %.1e' % (r['f32_mfma']['max_rel_err_vs_fp64'], r['bf16x6']['max_rel_err_vs_fp64'], r['bf16x9']['max_rel_err_vs_fp64']), r['bf16x6']['frac_of_bf16_peak'])"
