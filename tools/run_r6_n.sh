# Round 6, GPU session N: the round-5 tree (commit bb4fdf0, built under tools/build/r5tree) against this tree, same box, alternating bench runs
set -x
O=gpurun_out/r6n
mkdir -p $O
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  (cd $R/tools/build/r5tree && timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/bench_r5_$i.json 2> $R/$O/bench_r5_$i.err); echo "r5 run $i rc=$?"
  (cd $R && timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/bench_r6_$i.json 2> $R/$O/bench_r6_$i.err); echo "r6 run $i rc=$?"
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6n/bench_r*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ir=d['mfma_util'].get('in_timed_region',{})
        print(f.split('/')[-1], 'videos/s %.3f'%d['value'], 'decode TB/s %.3f'%d['decode_step']['tb_per_s'], 'vit %.4f prefill %.4f'%(d['mfma_util']['vit'], d['mfma_util']['prefill']), 'tower ms/video', ir.get('tower_ms_per_video'), 'single', d.get('single_video_latency_ms'), 'hbm', d['roofline_hbm']['frac'])
    except Exception as e: print(f, 'ERR', e)
P
