set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6f
mkdir -p $O
timeout 900 python tools/decode_b1_persistent_ab.py --arms 900,901,902,907,917,923,965,971 > $O/decode_b1_persistent_ab8.txt 2>&1; echo "b1 ab rc=$?"; grep -v amdgpu $O/decode_b1_persistent_ab8.txt | tail -9
timeout 600 python tools/decode_b1_repro.py --variants 900,907 --reps 3 --steps 200 > $O/b1_repro.txt 2>&1; echo "repro rc=$?"; grep -v amdgpu $O/b1_repro.txt | tail -2
