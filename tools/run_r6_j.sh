# Round 6, GPU session J: ViT attention knock-outs (what a persistent form could hide at most), then the default bench (decode defaults of this round)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6j
mkdir -p $O
timeout 600 python tools/attn_vit_big_probe.py --knockout > $O/attn_knockout.txt 2>&1; echo "attn ko rc=$?"; grep -v amdgpu $O/attn_knockout.txt | tail -12
timeout 1500 python bench.py --steps 4 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -c 3000 $O/bench_c2.json
