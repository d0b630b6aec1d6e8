# Round 6, GPU session V: decode batch / videos per step 128 vs 192 vs 256 on one box (pipelined c2)
set -x
python -c "from trace_amd import _lib; _lib.load()" || exit 9
O=gpurun_out/r6v
mkdir -p $O
for rep in 1 2; do
 for v in 128 256 192; do
  steps=3; [ $v -ge 192 ] && steps=2
  timeout 1200 python bench.py --videos-per-step $v --steps $steps --warmup 1 --no-cpu-baseline > $O/bench_v${v}_$rep.json 2> $O/bench_v${v}_$rep.err; echo "v=$v rep=$rep rc=$?"
 done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6v/bench_v*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'videos/s %.3f'%d['value'], 'decode TB/s %.3f'%d['decode_step']['tb_per_s'], 'gb/token %.3f'%d['decode_step']['gb_per_token'], d.get('stages_ms',{}).get('decode_ms_per_step'), d['mfma_util']['vit'], d['mfma_util']['prefill'], d.get('steps_repeat_exactly'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
P
