# Round 6, GPU session M: whole GPU suite on the tree with the new attention kernels + the default bench
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6m
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1500 python bench.py --steps 4 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r6m/bench_c2.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','decode_tok_s','single_video_latency_ms','steps_repeat_exactly']})
print(d['mfma_util']); print(d.get('decode_step')); print(d['roofline']['frac'], d['roofline_hbm']['frac'], d['roofline']['shapes'])
P
