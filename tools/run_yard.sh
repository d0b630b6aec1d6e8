python -c "from trace_amd import _lib; _lib.load()" || exit 1
timeout 300 python tools/gemm_lib_yardstick.py > gpurun_out/yard.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/yard_prof -o yard -- python /root/repo/tools/gemm_lib_yardstick.py > /dev/null 2>&1
cd /root/repo; ls gpurun_out/yard_prof | head
