# Round 6, GPU session H: workgroup target of the split-K partial GEMMs below 192 (804 = 128: o / down at 4 chunks; 803 = 96: qkv at 2), with / without write-through partial stores
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6h
mkdir -p $O
timeout 900 python tools/decode_variant_ab.py --batch 128 --steps 24 --rounds 5 --reset 740+705+808+144 \
  --variants 740+705+806,740+705+805,740+705+804,740+705+803,740+705+802,740+721+806,740+721+804,742+705+804 > $O/decode_ab.txt 2>&1; echo "decode ab rc=$?"; tail -10 $O/decode_ab.txt
