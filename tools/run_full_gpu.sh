# Round-3 measurement pass on one MI355X (gpurun): tests, smoke, bench lines for the three single-GPU configurations, rocprofv3 kernel
# statistics of the same commands, PMC traffic of the two roofline kernels.  Results land under gpurun_out/full3/ (scratch); the summaries
# quoted in DESIGN.md are copied to profiles/ by hand.  Run at the commit the round ends on (no kernel-source commit after it).
set -x
O=gpurun_out/full3
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# counter passes first: traffic.json is read by the bench lines below
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_kernels.py gemm attn_decode gemv > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_kernels.py gemm attn_decode gemv > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_sq -- python $R/tools/pmc_kernels.py gemm attn attn_decode > $R/$O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/traffic.json && cp $O/traffic.json profiles/traffic.json; head -8 $O/traffic.json
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; cat $O/bench_c2.json
python bench.py --steps 3 --warmup 1 --pipeline --no-cpu-baseline > $O/bench_c2_pipe.json 2> $O/bench_c2_pipe.err; echo "bench c2 pipelined rc=$?"
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python bench.py --config c5 --no-fp8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_bf16.json 2>> $O/bench_c5.err; echo "bench c5 bf16 rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/prof_bench.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec128 -- python $R/tools/decode_profile.py --batch 128 --steps 32 --eager > $R/$O/prof_dec128.log 2>&1; echo "prof dec128 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec1 -- python $R/tools/decode_profile.py --steps 32 --eager > $R/$O/prof_dec1.log 2>&1; echo "prof dec1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_vitstream -- python $R/tools/vit_stream_profile.py > $R/$O/prof_vitstream.log 2>&1; echo "prof vit stream rc=$?"
cd $R
for d in prof_bench prof_dec128 prof_dec1 prof_vitstream; do python tools/kernel_stats_top.py $O/$d 30 > $O/$d.top.txt; done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O
