# First GPU call of round 4 (≈ 15 GPU-minutes): what round 3 wrote after its GPU budget was spent and could not run.
#   1. the fp16 library's real-width tests (tests/test_gpu_f16.py behind TRACE_TEST_F16_WIDE=1) -> if green, drop the gate
#   2. branch r4-ticket-base: check it out FIRST (git checkout r4-ticket-base && python -m trace_amd.build), then run this script's stress part:
#      with the round-3 kernel 3 of 109 pipelined steps differ (profiles/r03_pipeline_stress_stages.txt); the branch must show none, twice
#   3. tools/gemm_pers_ab.py on the branch: no timing change expected
set -x
O=gpurun_out/r4first
mkdir -p $O
TRACE_TEST_F16_WIDE=1 timeout 600 python -m pytest tests/test_gpu_f16.py -q --tb=short -p no:cacheprovider > $O/f16_wide.log 2>&1; echo "f16 wide rc=$?"; tail -5 $O/f16_wide.log
if git rev-parse --abbrev-ref HEAD 2>/dev/null | grep -q r4-ticket-base || grep -q "TicketBase" trace_amd/csrc/gemm_pers.hip; then
  timeout 120 python -m pytest tests/test_gpu_kernels.py -k "persistent or layernorm_fold" -q -p no:cacheprovider > $O/pers_tests.log 2>&1; echo "pers tests rc=$?"
  timeout 420 python tools/pipeline_stress.py --steps 110 > $O/stress_a.txt 2>&1; tail -4 $O/stress_a.txt | cut -c1-300
  timeout 420 python tools/pipeline_stress.py --steps 110 > $O/stress_b.txt 2>&1; tail -4 $O/stress_b.txt | cut -c1-300
  timeout 300 python tools/gemm_pers_ab.py > $O/gemm_pers_ab.txt 2>&1; tail -12 $O/gemm_pers_ab.txt
fi
