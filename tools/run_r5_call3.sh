set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r5c3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "attention" > $O/attn_tests.log 2>&1; echo "attention tests rc=$?"; tail -8 $O/attn_tests.log
timeout 200 python tools/attn_vit_big_probe.py > $O/attn_big.txt 2>&1; cat $O/attn_big.txt
timeout 200 python tools/attn_vit_big_probe.py --knockout > $O/attn_knockout.txt 2>&1; cat $O/attn_knockout.txt
