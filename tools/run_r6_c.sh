set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6c
mkdir -p $O
B=740+705+806+770
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 740+705+808+762+770+100+780 \
  --variants $B+760+780+100,$B+760+786+100,$B+760+790+100,$B+763+780+100,$B+764+780+100,$B+763+780+108,$B+764+780+108,$B+760+786+108 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -10 $O/decode_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -5 $O/pytest_parity.log
