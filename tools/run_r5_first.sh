# First GPU call of the next round (~25 GPU-minutes): what round 4 prepared and could not afford.
#   1. the LayerNorm-fold path under the per-layer checksum tracer until it catches a wrong row panel: 2 x 160 pipelined steps, fold on (151).
#      The tool prints, for every differing step, the FIRST (tower call, layer, stage, 256-row panel) whose output differs from step 0's and how
#      the difference spreads through the stages that follow: producer epilogue (stage 2 / 5 differ first), finalize (3 / 6 first) or consumer
#      (0 / 4 first while 3 / 6 agree).  DESIGN 5a.r4 lists what is already excluded.
#   2. the wide decode step's split-K partition: TRACE_PARTIAL_WGS = 128 / 256 (shipped) / 512 on tools/decode_profile.py --batch 128.
set -x
O=gpurun_out/r5first
mkdir -p $O
timeout 900 python tools/pipeline_stress.py --steps 160 --max-new 200 --plan 151,151 --trace > $O/stress_trace.txt 2>&1; echo "trace rc=$?"; grep -v "^  step [0-9]*:" $O/stress_trace.txt | tail -12 | cut -c1-700
for w in 128 256 512; do TRACE_PARTIAL_WGS=$w timeout 200 python tools/decode_profile.py --batch 128 --steps 48 > $O/decode_b128_wgs$w.txt 2>&1; tail -1 $O/decode_b128_wgs$w.txt; done
