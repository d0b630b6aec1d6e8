# Round 5, GPU call 1: the three kernel changes of the round, each first for correctness, then timed against what it replaces.
#   (1) kernel tests: register-staged GEMM loaders (332), the 192-row ViT attention (191 / 192), the rest of test_gpu_kernels
#   (2) interleaved A/Bs: GEMM family on the ViT / prefill shapes; ViT attention; decode touch-prefetch at batch 128 and batch 1
set -x
O=gpurun_out/r5c1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --tb=short -p no:cacheprovider > $O/kernels.log 2>&1; echo "kernel tests rc=$?"; tail -15 $O/kernels.log
timeout 300 python tools/attn_vit_big_probe.py > $O/attn_big.txt 2>&1; echo "attn probe rc=$?"; cat $O/attn_big.txt
timeout 600 python tools/gemm_pers_ab.py 32 vit prefill > $O/gemm_ab.txt 2>&1; echo "gemm ab rc=$?"; cat $O/gemm_ab.txt
timeout 400 python tools/decode_variant_ab.py --batch 128 --variants 600,616,632,664 > $O/dec128_prefetch.txt 2>&1; echo "dec128 rc=$?"; tail -6 $O/dec128_prefetch.txt
timeout 400 python tools/decode_variant_ab.py --batch 1 --steps 48 --variants 600,632,664,696 > $O/dec1_prefetch.txt 2>&1; echo "dec1 rc=$?"; tail -6 $O/dec1_prefetch.txt
for w in 128 512; do TRACE_PARTIAL_WGS=$w timeout 200 python tools/decode_profile.py --batch 128 --steps 48 --eager > $O/decode_b128_wgs$w.txt 2>&1; tail -1 $O/decode_b128_wgs$w.txt; done
