# Round 6, session E: the whole GPU suite + smoke on the current tree, then the c2 bench line
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6e
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest.log
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-600 $O/bench_c2.json
