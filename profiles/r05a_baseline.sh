# profiles/r05a_baseline.sh -- where finalize's wall time goes (MSNV_FINALIZE_TRACE marks) and the HIP API split of one dataset build
export TMPDIR=/tmp
MSNV_FINALIZE_TRACE=1 python3 profiles/pack_prof.py > gpurun_out/r05a_pack_prof.json 2> gpurun_out/r05a_finalize_trace.txt
rm -rf gpurun_out/r05a_prof
rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats -d gpurun_out/r05a_prof -o p --output-format csv -- python3 profiles/pack_prof.py > gpurun_out/r05a_prof.log 2>&1
rm -f gpurun_out/r05a_prof/p_hip_api_trace.csv gpurun_out/r05a_prof/p_kernel_trace.csv
