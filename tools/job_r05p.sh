# code-size experiment on the small-lattice kernels: the same source under four optimisation settings, C1 / C2 lines of each
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R/libdmet_preview_amd/csrc
BASE="-std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function"
for V in O3 Os O3nounroll Oz O3; do
  case $V in
    O3) F="-O3";; Os) F="-Os";; O3nounroll) F="-O3 -fno-unroll-loops";; Oz) F="-Oz";;
  esac
  /opt/rocm/bin/hipcc $F $BASE -c small.hip -o small.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libdmetk.so || exit 1
  for WL in C1 C2; do
    (cd $R && python bench.py --workload $WL --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_${WL}_$V.json 2> $O/bench_${WL}_$V.err)
    python3 - <<PY
import json
d=json.loads(open("$O/bench_${WL}_$V.json").read().strip().splitlines()[-1])
print("RESULT $V $WL ms_per_step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["roofline"]["stages"].items()}, "parity", d.get("parity_stages_ok"))
PY
  done
done 2>&1 | tee $O/log.txt
cd $R && python -m pytest tests/test_gpu_small.py -m gpu -x -q 2>&1 | tail -3
grep RESULT $O/log.txt
