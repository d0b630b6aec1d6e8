# Round 6, GPU session R: regression of the decode paths after the last-block V^T group zeroing (kernel tests, parity, full-size batch tests, drop-in)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6r
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py tests/test_gpu_f16.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 4 --variants 0 --reset 0 2>&1 | grep "^batch"
