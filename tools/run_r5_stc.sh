set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r5stc
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stc.py -q --tb=short -p no:cacheprovider -s > $O/stc_tests.log 2>&1; echo "stc tests rc=$?"; grep -E "passed|failed|STC at real" $O/stc_tests.log
python bench.py --config stc --steps 10 --warmup 2 > $O/bench_stc.json 2> $O/bench_stc.err; echo "bench stc rc=$?"; cat $O/bench_stc.json | cut -c1-900
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stc -- python $R/bench.py --config stc --steps 5 --warmup 1 > $R/$O/prof_stc.log 2>&1; echo "prof stc rc=$?"
cd $R
python tools/kernel_stats_top.py $O/prof_stc 16
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
