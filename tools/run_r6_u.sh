# Round 6, GPU session U: prefill runs of up to 8 equal-length prompts (adaptive group): parity / full-size / drop-in suites, then c4 and c2 benches
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6u
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py tests/test_gpu_f16.py tests/test_gpu_fp8.py -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 900 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python - <<'P'
import json
for f in ('bench_c4','bench_c2'):
    d=json.loads(open('gpurun_out/r6u/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'videos/s %.3f'%d['value'], d['mfma_util']['vit'], d['mfma_util']['prefill'], d.get('stages_ms'), d.get('steps_repeat_exactly'))
P
