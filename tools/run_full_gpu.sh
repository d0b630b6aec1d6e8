set -x
mkdir -p gpurun_out/full
timeout 1200 python -m pytest tests -m gpu -q --tb=short --timeout=600 -p no:cacheprovider > gpurun_out/full/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/full/pytest.log
python __graft_entry__.py --smoke > gpurun_out/full/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/full/smoke.log
python bench.py --steps 2 --warmup 1 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$?"; cat gpurun_out/full/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/full/prof_bench.log 2>&1; echo "prof rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/prof_dec -- python $GRAFT_REPO_ROOT/tools/decode_profile.py --steps 32 --eager > $GRAFT_REPO_ROOT/gpurun_out/full/prof_dec.log 2>&1; echo "prof dec rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/prof_vit -- python $GRAFT_REPO_ROOT/tools/vit_profile.py > $GRAFT_REPO_ROOT/gpurun_out/full/prof_vit.log 2>&1; echo "prof vit rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/prof_vitstream -- python $GRAFT_REPO_ROOT/tools/vit_stream_profile.py > $GRAFT_REPO_ROOT/gpurun_out/full/prof_vitstream.log 2>&1; echo "prof vit stream rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py gemm gemv > $GRAFT_REPO_ROOT/gpurun_out/full/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/full/pmc_write -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py gemm gemv > $GRAFT_REPO_ROOT/gpurun_out/full/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT; python tools/pmc_traffic.py gpurun_out/full/pmc_fetch gpurun_out/full/pmc_write > gpurun_out/full/traffic.json; cat gpurun_out/full/traffic.json | head -5
find gpurun_out/full -name '*kernel_trace.csv' -delete; find gpurun_out/full -name '*.csv' | head
