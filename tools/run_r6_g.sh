# Round 6, GPU session G: the wide step without qkv_finish (144/145: the attention's prologue sums the qkv partial rows), on top of the split-K knobs; full GPU suite after the
# persistent-kernel removal; kernel statistics of the wide decode step.
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6g
mkdir -p $O
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 740+705+808+144 \
  --variants 740+705+808+144,740+705+808+145,740+705+806+144,740+705+806+145,742+705+806+144,742+705+806+145,741+705+806+145,743+705+806+145 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -10 $O/decode_ab.txt
timeout 600 python tools/decode_variant_ab.py --batch 64 --steps 24 --rounds 4 --reset 740+705+808+144 \
  --variants 740+705+808+144,740+705+806+145,742+705+806+145 > $O/decode_ab64.txt 2>&1; echo "decode ab64 rc=$?"; tail -4 $O/decode_ab64.txt
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
