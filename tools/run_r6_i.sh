# Round 6, GPU session I: context splits of the decode attention at batch 128 (1024 workgroups of 4 waves over 768 resident slots = 1.33 rounds; 3 splits = 4 whole rounds)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6i
mkdir -p $O
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 850 \
  --variants 850,852,853,854,856,859,853+763,852+763 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -10 $O/decode_ab.txt
timeout 600 python tools/decode_variant_ab.py --batch 64 --steps 24 --rounds 4 --reset 850 \
  --variants 850,852,853,856 > $O/decode_ab64.txt 2>&1; echo "decode ab64 rc=$?"; tail -5 $O/decode_ab64.txt
