# Round 6 measurement pass on one MI355X (gpurun): the whole GPU suite, smoke, counter passes (traffic of the roofline kernels and of the decode GEMMs; SQ
# counters of the tower's kernels and both attention kernels), bench lines of every configuration (+ the STC connector), rocprofv3 kernel statistics of the tower /
# decode / bench commands, and a pipelined stress run that adds to the count of clean steps.  Results land under gpurun_out/full6/ (scratch); the summaries quoted in
# DESIGN.md are copied to profiles/ by hand.  (The single-purpose A/B calls of the round quote their command lines in the header of the profiles/ file they produced.)
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/full6
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
timeout 1800 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest.log
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python $R/tools/pmc_kernels.py gemm attn_decode gemv decgemm > $R/$O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python $R/tools/pmc_kernels.py gemm attn_decode gemv decgemm > $R/$O/pmc_write.log 2>&1; echo "pmc write rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_sq -- python $R/tools/pmc_kernels.py tower attn_decode > $R/$O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_insts -- python $R/tools/pmc_kernels.py tower > $R/$O/pmc_insts.log 2>&1; echo "pmc insts rc=$?"
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/traffic.json && cp $O/traffic.json profiles/traffic.json; head -6 $O/traffic.json
python tools/pmc_summary.py $O/pmc_sq $O/pmc_insts > $O/pmc_sq.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt
python bench.py --gpus 1 --steps 6 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-1500 $O/bench_c2.json
python bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline > $O/bench_c2_seq.json 2> $O/bench_c2_seq.err; echo "bench c2 sequential rc=$?"
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python bench.py --config stc --steps 10 --warmup 2 > $O/bench_stc.json 2> $O/bench_stc.err; echo "bench stc rc=$?"; cat $O/bench_stc.json
python bench.py --config c1 > $O/bench_c1.json 2> $O/bench_c1.err; echo "bench c1 rc=$?"; cut -c1-600 $O/bench_c1.json
timeout 400 python tools/pipeline_stress.py --steps 60 --max-new 200 --plan 0 > $O/pipeline_stress.txt 2>&1; echo "stress rc=$?"; tail -3 $O/pipeline_stress.txt
timeout 300 python tools/gemm_w4_check.py --time > $O/gemm_w4_check.txt 2>&1; echo "w4 check rc=$?"; grep -c "^ok" $O/gemm_w4_check.txt; grep "mismatching" $O/gemm_w4_check.txt
timeout 300 python tools/attn_vit_big_probe.py > $O/attn_probe.txt 2>&1; echo "attn probe rc=$?"; grep -v amdgpu $O/attn_probe.txt | tail -6
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_vitstream -- python $R/tools/vit_stream_profile.py > $R/$O/prof_vitstream.log 2>&1; echo "prof vit stream rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec128 -- python $R/tools/decode_profile.py --batch 128 --steps 32 --eager > $R/$O/prof_dec128.log 2>&1; echo "prof dec128 rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec1 -- python $R/tools/decode_profile.py --steps 32 --eager > $R/$O/prof_dec1.log 2>&1; echo "prof dec1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline > $R/$O/prof_bench.log 2>&1; echo "prof bench rc=$?"
cd $R
for d in prof_vitstream prof_dec128 prof_dec1 prof_bench; do python tools/kernel_stats_top.py $O/$d 30 > $O/$d.top.txt; done
head -14 $O/prof_vitstream.top.txt; head -12 $O/prof_dec128.top.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O
