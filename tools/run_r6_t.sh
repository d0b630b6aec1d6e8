set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6t
mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style bench rc=$?"
timeout 900 python bench.py --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_fp16.json 2> $O/bench_c2_fp16.err; echo "fp16 bench rc=$?"
timeout 1800 python tools/pipeline_stress.py --steps 300 --max-new 200 --plan 0 > $O/pipeline_stress300.txt 2>&1; echo "stress rc=$?"; tail -3 $O/pipeline_stress300.txt
python - <<'P'
import json
for f in ('bench_driver_style','bench_c2_fp16'):
    d=json.loads(open('gpurun_out/r6t/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'videos/s %.3f'%d['value'], 'ms/step %.1f'%d['ms_per_step'], d['mfma_util']['vit'], d['mfma_util']['prefill'], d['roofline']['frac'], d['roofline_hbm']['frac'], d.get('steps_repeat_exactly'), d.get('single_video_latency_ms'))
P
