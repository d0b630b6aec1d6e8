# Round 4, GPU call 3 (the last): the shipped configuration — LayerNorm fold OFF, pipeline ON.  (1) the stress tool on it at high exposure (decode stage as
# long as the encode stage), 110 steps; (2) the whole GPU suite; (3) a 12-step pipelined C2 line + C4 / C5; (4) rocprofv3 kernel statistics of the tower
# call shape and the PMC traffic passes of the shipped fc1 kernel.  Results under gpurun_out/r4c3/.
set -x
O=gpurun_out/r4c3
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
R=$GRAFT_REPO_ROOT
timeout 900 python tools/pipeline_stress.py --steps 110 --max-new 200 --plan 0 > $O/stress_shipped.txt 2>&1; echo "stress rc=$?"; grep -v "^  step" $O/stress_shipped.txt | tail -4 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
python bench.py --steps 12 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-300 $O/bench_c2.json
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"; cut -c1-200 $O/bench_c4.json
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"; cut -c1-200 $O/bench_c5_fp8.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_vitstream -- python $R/tools/vit_stream_profile.py > $R/$O/prof_vitstream.log 2>&1; echo "prof vit stream rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_kernels.py gemm attn_decode gemv > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_kernels.py gemm attn_decode gemv > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/traffic.json; head -30 $O/traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt
python tools/kernel_stats_top.py $O/prof_vitstream 30 > $O/prof_vitstream.top.txt 2>/dev/null; f=$(find $O/prof_vitstream -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/prof_vitstream.kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O
