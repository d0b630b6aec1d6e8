# Round 6, GPU session A (one MI355X): cheap measurements that decide where the round's kernel work goes.
#   1. wide decode step, interleaved A/B of the split-K GEMM knobs (nt weight DMA, nt / write-through partial stores, 128 x 64 tiles, workgroup targets)
#   2. bench c2, pipelined: eager vs hipGraph replay, then graph again (three runs on one box: do mfma_util.vit / .prefill agree within 2 %?)
#   3. the co-residency probe
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6a
mkdir -p $O
timeout 600 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 740+705+808 \
  --variants 740+705+808,740+707+808,740+713+808,740+721+808,740+729+808,741+705+808,742+705+808,743+705+808,741+705+806,743+721+808,740+705+806 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -14 $O/decode_ab.txt
timeout 300 python tools/coresidency_probe.py > $O/coresidency_probe.txt 2> $O/coresidency_probe.err; echo "probe rc=$?"; cat $O/coresidency_probe.txt; tail -3 $O/coresidency_probe.err
timeout 900 python bench.py --steps 3 --warmup 1 --eager > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?"; tail -3 $O/bench_eager.err
timeout 900 python bench.py --steps 3 --warmup 1 --graph --no-cpu-baseline > $O/bench_graph1.json 2> $O/bench_graph1.err; echo "bench graph rc=$?"; tail -3 $O/bench_graph1.err
timeout 900 python bench.py --steps 3 --warmup 1 --graph --no-cpu-baseline > $O/bench_graph2.json 2> $O/bench_graph2.err; echo "bench graph rc=$?"
python - <<'P'
import json
for f in ("bench_eager", "bench_graph1", "bench_graph2"):
    try:
        j = json.load(open(f"gpurun_out/r6a/{f}.json"))
        print(f, "videos/s %.3f" % j["value"], "dec ms/step %.3f" % j["stages_ms"]["decode_ms_per_step"], "mfma", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j["mfma_util"].items() if k in ("vit", "prefill", "vit_min_max", "prefill_min_max")},
              "in_region", {k: round(v, 4) for k, v in (j["mfma_util"]["in_timed_region"] or {}).items() if isinstance(v, float)}, "one", j["single_video_latency_ms"], "hbm", j["roofline_hbm"]["frac"], "repeat", j["steps_repeat_exactly"])
    except Exception as e:
        print(f, "unreadable:", e)
P
