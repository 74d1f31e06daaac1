set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-parity --no-full-config --steps 1 --warmup 0 --fit-iters 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof_trace -- $B > $R/gpurun_out/r03_prof_trace.json 2> $R/gpurun_out/r03_prof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r03_prof_fetch -- $B > /dev/null 2> $R/gpurun_out/r03_prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r03_prof_write -- $B > /dev/null 2> $R/gpurun_out/r03_prof_write.err
ls $R/gpurun_out/r03_prof_*/*/ 
