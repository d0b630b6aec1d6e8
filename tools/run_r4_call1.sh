# Round 4, GPU call 1: (1) the fp16 library's real-width tests that never ran on hardware; (2) the pipelined schedule's rare wrong ViT tile,
# de-confounded: two 110-step phases on the shipped tile walk (tickets, atomic re-arm), one on round 3's plain-store re-arm (positive control);
# if the shipped walk still differs, a static-deal phase (fold on) and a no-fold phase (tickets on) name the culprit; (3) a short pipelined bench.
set -x
O=gpurun_out/r4c1
mkdir -p $O
TRACE_TEST_F16_WIDE=1 timeout 600 python -m pytest tests/test_gpu_f16.py -q --tb=short -p no:cacheprovider > $O/f16_wide.log 2>&1; echo "f16 wide rc=$?"; tail -5 $O/f16_wide.log
timeout 120 python -m pytest tests/test_gpu_kernels.py -k "persistent or layernorm_fold" -q -p no:cacheprovider > $O/pers_tests.log 2>&1; echo "pers tests rc=$?"; tail -3 $O/pers_tests.log
timeout 2100 python tools/pipeline_stress.py --steps 110 --plan 500,500,502 --adaptive > $O/stress.txt 2>&1; echo "stress rc=$?"; grep -v "^  step" $O/stress.txt | tail -12 | cut -c1-300
timeout 400 python bench.py --steps 3 --warmup 1 --pipeline > $O/bench_pipe3.json 2> $O/bench_pipe3.err; echo "bench rc=$?"; cut -c1-600 $O/bench_pipe3.json
