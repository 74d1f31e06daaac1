# rocprofv3 passes behind profiles/rNN_*: per workload one --kernel-trace --stats pass and three --pmc passes (FETCH_SIZE,
# WRITE_SIZE, wave states), each in its OWN run (gpurun refuses --pmc combined with the other trace domains).
# usage (on the GPU box): bash tools/prof_passes.sh <tag>      -> gpurun_out/<tag>/{C5,C4}_{trace,fetch,write,waves}/ + summaries
set -x
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
for WL in C5 C4; do
  if [ $WL = C5 ]; then EXTRA="--scaling weak --kl-per-gpu 14 --no-full-config"; else EXTRA="--workload C4"; fi
  B="python3 $R/bench.py $EXTRA --no-cpu-baseline --no-parity --steps 1 --warmup 0 --fit-iters 0"
  rocprofv3 --kernel-trace --stats -d $O/${WL}_trace -- $B > $O/${WL}_trace.json 2> $O/${WL}_trace.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${WL}_fetch -- $B > /dev/null 2> $O/${WL}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${WL}_write -- $B > /dev/null 2> $O/${WL}_write.err
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${WL}_waves -- $B > /dev/null 2> $O/${WL}_waves.err
done
cd $R
for WL in C5 C4; do
  T=$(ls $O/${WL}_trace/*/*.db | head -1); F=$(ls $O/${WL}_fetch/*/*.db | head -1); W=$(ls $O/${WL}_write/*/*.db | head -1)
  python3 tools/rocprof_summary.py $T $F $W > $O/${WL}_kernel_trace_and_pmc_summary.txt
  python3 tools/rocprof_summary.py --traffic-json $O/traffic_latest.json --workload $WL $F $W
  python3 tools/pmc_wave_states.py $O/${WL}_waves > $O/${WL}_pmc_wave_states.txt
done
rm -rf $O/*_trace $O/*_fetch $O/*_write $O/*_waves
ls -la $O
