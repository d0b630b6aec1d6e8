set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short --timeout=300 -p no:cacheprovider -x -k "persistent" > $O/pytest_b1.log 2>&1; echo "pytest b1 rc=$?"; tail -15 $O/pytest_b1.log
timeout 600 python tools/decode_b1_persistent_ab.py > $O/decode_b1_persistent_ab.txt 2>&1; echo "b1 ab rc=$?"; grep -v amdgpu $O/decode_b1_persistent_ab.txt | tail -12
