# Round 5, GPU call 2: the 192-row attention after the hazard fix (determinism), knock-out runs and SQ counters of both attention kernels, and the
# persistent GEMM's tile walk (tickets vs static deal) with and without the A-panel touches.
set -x
python -c "from trace_amd import _lib; _lib.load(); _lib.load('f16')" || exit 9
O=gpurun_out/r5c2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x --tb=short -p no:cacheprovider -k "attention" > $O/attn_tests.log 2>&1; echo "attention tests rc=$?"; tail -5 $O/attn_tests.log
timeout 200 python tools/attn_vit_big_probe.py > $O/attn_big.txt 2>&1; cat $O/attn_big.txt
timeout 200 python tools/attn_vit_big_probe.py --knockout > $O/attn_knockout.txt 2>&1; cat $O/attn_knockout.txt
timeout 400 python tools/gemm_pers_ab.py 0 8 vit > $O/gemm_walk_ab.txt 2>&1; cat $O/gemm_walk_ab.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$O/counters_list.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_a -- python $R/tools/attn_vit_pmc.py 64 > $R/$O/pmc_a.log 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_b -- python $R/tools/attn_vit_pmc.py 64 > $R/$O/pmc_b.log 2>&1; echo "pmc b rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_c -- python $R/tools/attn_vit_pmc.py 64 > $R/$O/pmc_c.log 2>&1; echo "pmc c rc=$?"
cd $R
python tools/pmc_summary.py $O/pmc_a $O/pmc_b $O/pmc_c --match=attn_vit > $O/pmc_attn.txt; cat $O/pmc_attn.txt
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
