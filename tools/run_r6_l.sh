# Round 6, GPU session L: decode attention with unconditional cache loads / a two-part block loop / the prologue loads waited for before the loop (counted waits in the
# steady loop instead of vmcnt(0)) against the round-5 form (772)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6l
mkdir -p $O
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 770 \
  --variants 770,772,770+763,771 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -6 $O/decode_ab.txt
timeout 600 python tools/decode_variant_ab.py --batch 1 --steps 48 --rounds 5 --reset 770 --variants 770,772 > $O/decode_ab1.txt 2>&1; echo "decode ab1 rc=$?"; tail -3 $O/decode_ab1.txt
timeout 600 python tools/decode_variant_ab.py --batch 16 --steps 32 --rounds 5 --reset 770 --variants 770,772 > $O/decode_ab16.txt 2>&1; echo "decode ab16 rc=$?"; tail -3 $O/decode_ab16.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
