# Round 6, GPU session O: the event-guarded staging ring (no stream drain in encode_tail / splice) against HEAD (built under tools/build/headtree), same box, alternating
set -x
O=gpurun_out/r6o
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for i in 1 2; do
  (cd $R/tools/build/headtree && timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/bench_head_$i.json 2> $R/$O/bench_head_$i.err); echo "head run $i rc=$?"
  (cd $R && timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/bench_new_$i.json 2> $R/$O/bench_new_$i.err); echo "new run $i rc=$?"
done
(cd $R/tools/build/headtree && timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $R/$O/bench_head_seq.json 2> $R/$O/bench_head_seq.err); echo "head seq rc=$?"
(cd $R && timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $R/$O/bench_new_seq.json 2> $R/$O/bench_new_seq.err); echo "new seq rc=$?"
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6o/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'videos/s %.3f'%d['value'], 'decode TB/s %.3f'%d['decode_step']['tb_per_s'], 'vit %.4f prefill %.4f'%(d['mfma_util']['vit'], d['mfma_util']['prefill']), 'single', d.get('single_video_latency_ms'), d.get('steps_repeat_exactly'))
    except Exception as e: print(f, 'ERR', e)
P
