set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6d
mkdir -p $O
timeout 900 python tools/decode_b1_persistent_ab.py --arms 900,901,906,907 > $O/decode_b1_persistent_ab4.txt 2>&1; echo "b1 ab rc=$?"; grep -v amdgpu $O/decode_b1_persistent_ab4.txt | tail -5
