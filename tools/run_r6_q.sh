# Round 6, GPU session Q: the new three-distinct-frames ViT parity test, then 200 pipelined stress steps on the final tree (new attention kernels, staging ring)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r6q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short --timeout=900 -p no:cacheprovider -k "large_geometry" -s > $O/pytest_vit.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu $O/pytest_vit.log | tail -12
timeout 1500 python tools/pipeline_stress.py --steps 200 --max-new 200 --plan 0 > $O/pipeline_stress200.txt 2>&1; echo "stress rc=$?"; tail -3 $O/pipeline_stress200.txt
