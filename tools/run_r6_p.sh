# Round 6, GPU session P: the decode GEMV's weight loop with counted waits (whole batches unconditional, partial batch apart) against HEAD (tools/build/headtree), one box
set -x
O=gpurun_out/r6p
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q --tb=short --timeout=900 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for B in 1 4 16; do
 for i in 1 2; do
  (cd $R/tools/build/headtree && timeout 600 python tools/decode_variant_ab.py --batch $B --steps 48 --rounds 5 --variants 0 --reset 0 2>&1 | grep "^batch" | sed 's/^/HEAD /') >> $O/gemv_ab.txt
  (cd $R && timeout 600 python tools/decode_variant_ab.py --batch $B --steps 48 --rounds 5 --variants 0 --reset 0 2>&1 | grep "^batch" | sed 's/^/NEW  /') >> $O/gemv_ab.txt
 done
done
cat $O/gemv_ab.txt
