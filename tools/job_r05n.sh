# round 5 evidence pass: rocprofv3 trace + PMC passes of C5 (14-kL shard) and C4, then the driver's command line, then the C1-C4 lines
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05n
mkdir -p $O
bash $R/tools/prof_passes.sh r05n > $O/prof_passes.log 2>&1
cd $R
timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
for WL in C1 C2 C3 C4; do timeout 600 python bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err; echo $WL rc=$?; done
tail -c 600 $O/bench_default.json
