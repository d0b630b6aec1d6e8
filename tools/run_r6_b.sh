# Round 6, GPU session B: decode attention variants (3-wave workgroups, nt loads, blocked-V knock-out), split-K shapes on top of the 192-workgroup target,
# the prefill last-rows shortcut (bit-identity tests + timing), kernel statistics of the wide decode step.
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6b
mkdir -p $O
B=740+705+806
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 740+705+808+762+770+100 \
  --variants $B+760+770+100,$B+761+770+100,$B+760+771+100,$B+761+771+100,$B+760+770+108,$B+761+771+108,742+705+806+760+770+100,741+721+806+760+770+100,740+705+808+760+770+100 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -12 $O/decode_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -5 $O/pytest_parity.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --tb=short --timeout=900 -p no:cacheprovider -x -k "last_rows or batch_128 or graph_equals" > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -5 $O/pytest_full.log
python - <<'P' > $O/prefill_shortcut_ab.txt 2>&1
# prefill of four 1967-row prompts: last decoder layer over all rows (750) vs over the last rows (751), interleaved
import statistics, sys, torch
sys.path.insert(0, ".")
from trace_amd import config as tcfg, synth
from trace_amd.engine import TraceEngine, ops
cfg = tcfg.trace_7b()
eng = TraceEngine(cfg, max_batch=4, max_ctx=2048, max_frames=128, max_new_tokens=8)
eng.load_weights(synth.iter_weights(cfg, device="cuda"))
embs = [(torch.randn(1967, cfg.hidden_size, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(4)]
ts = {0: [], 1: []}
for r in range(7):
    for mode in (0, 1):
        ops.set_gemm_variant(750 + mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.prefill_multi(0, embs); b.record(); torch.cuda.synchronize()
        ts[mode].append(a.elapsed_time(b) / 4)
ops.set_gemm_variant(751)
for mode in (0, 1):
    print(f"prefill per prompt (run of four, L = 1967), last layer over {'the last rows only' if mode else 'all rows'}: median {statistics.median(ts[mode][1:]):.3f} ms  ({' '.join('%.3f' % t for t in ts[mode])})")
P
cat $O/prefill_shortcut_ab.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec128 -- python $R/tools/decode_profile.py --batch 128 --steps 32 --eager > $R/$O/prof_dec128.log 2>&1; echo "prof dec128 rc=$?"
cd $R
python tools/kernel_stats_top.py $O/prof_dec128 30 > $O/prof_dec128.top.txt; head -20 $O/prof_dec128.top.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
