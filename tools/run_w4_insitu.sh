python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/gemm_w4_check.py --time > gpurun_out/w4_final_check.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/w4_tests.txt 2>&1; tail -3 gpurun_out/w4_tests.txt
for v in 0 1 0 1; do
  TRACE_GEMM_W4=$v timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/w4_bench_$v.json.tmp 2> gpurun_out/w4_bench_err.txt && cat gpurun_out/w4_bench_$v.json.tmp >> gpurun_out/w4_bench_$v.json
done
tail -2 gpurun_out/w4_bench_err.txt
