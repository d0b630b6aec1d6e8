# Round 6, GPU session K: the 192-row ViT attention with the trailing key's loads hoisted, 16-byte output stores and buffer-descriptor LDS-DMA (no scratch reload in the tile loop)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6k
mkdir -p $O
timeout 600 python tools/attn_vit_big_probe.py > $O/attn_probe.txt 2>&1; echo "attn probe rc=$?"; grep -v amdgpu $O/attn_probe.txt | tail -7
timeout 600 python tools/attn_vit_big_probe.py --knockout > $O/attn_knockout.txt 2>&1; echo "attn ko rc=$?"; grep -v amdgpu $O/attn_knockout.txt | tail -6
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_f16.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
