# Round 4, GPU call 2: the whole GPU suite on the round's kernels (fused ViT front end, SwiGLU-fold batch-1 step, atomic ticket re-arm), the bench
# lines of every configuration, the A/Bs that decide the defaults (pipeline on / off, 128 / 256 sequences per decode batch, SwiGLU fold), and the
# rocprofv3 kernel statistics the DESIGN tables quote.  Results under gpurun_out/r4c2/ (scratch); summaries are copied to profiles/ by hand.
set -x
O=gpurun_out/r4c2
mkdir -p $O
rm -f gpurun_out/parity_measured.txt
# (0) call 1 showed the wrong ViT tile with the STATIC tile deal too and never without the LayerNorm fold: not the tickets.  The fold's row statistics now
# move with agent-scope atomics; three phases at 8x the exposure per step (decode stage as long as the encode stage): shipped / round-3 plain accesses / shipped
timeout 1500 python tools/pipeline_stress.py --steps 70 --max-new 200 --plan 510,511,510 > $O/stress_stats.txt 2>&1; echo "stress rc=$?"; grep -v "^  step" $O/stress_stats.txt | tail -8 | cut -c1-300
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
cp gpurun_out/parity_measured.txt $O/ 2>/dev/null
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py --steps 4 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"; cut -c1-400 $O/bench_c2.json
python bench.py --steps 4 --warmup 1 --no-pipeline --no-cpu-baseline > $O/bench_c2_seq.json 2> $O/bench_c2_seq.err; echo "bench c2 sequential rc=$?"; cut -c1-200 $O/bench_c2_seq.json
python bench.py --steps 3 --warmup 1 --videos-per-step 256 --no-cpu-baseline > $O/bench_c2_b256.json 2> $O/bench_c2_b256.err; echo "bench c2 b256 rc=$?"; cut -c1-200 $O/bench_c2_b256.json
python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"; cut -c1-200 $O/bench_c4.json
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_fp8.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"; cut -c1-200 $O/bench_c5_fp8.json
python bench.py --config c5 --no-fp8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_bf16.json 2>> $O/bench_c5.err; echo "bench c5 bf16 rc=$?"
python bench.py --config c1 --steps 5 --warmup 2 > $O/bench_c1.json 2> $O/bench_c1.err; echo "bench c1 rc=$?"; cut -c1-300 $O/bench_c1.json
timeout 300 python tools/decode_b1_ab.py > $O/decode_b1_ab.txt 2>&1; tail -4 $O/decode_b1_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_vitstream -- python $R/tools/vit_stream_profile.py > $R/$O/prof_vitstream.log 2>&1; echo "prof vit stream rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec1 -- python $R/tools/decode_profile.py --steps 32 --eager > $R/$O/prof_dec1.log 2>&1; echo "prof dec1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec128 -- python $R/tools/decode_profile.py --batch 128 --steps 32 --eager > $R/$O/prof_dec128.log 2>&1; echo "prof dec128 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline > $R/$O/prof_bench.log 2>&1; echo "prof bench rc=$?"
cd $R
for d in prof_vitstream prof_dec1 prof_dec128 prof_bench; do python tools/kernel_stats_top.py $O/$d 30 > $O/$d.top.txt 2>/dev/null; f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/$d.kernel_stats.csv; done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O
