# round-5 differential campaigns (new seeds) after the round's kernel changes: small-lattice kernels (meanfield / iteration stress
# hit them whenever a random lattice fits their limits), batched generator, fused fit objective, split-K small products
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
STRESS_SEED=505 STRESS_TRIALS=250 timeout 900 python tools/meanfield_stress.py > $O/meanfield_stress.txt 2>&1; tail -4 $O/meanfield_stress.txt
STRESS_SEED=506 STRESS_TRIALS=120 timeout 900 python tools/iteration_stress.py > $O/iteration_stress.txt 2>&1; tail -4 $O/iteration_stress.txt
STRESS_SEED=507 STRESS_TRIALS=150 timeout 900 python tools/fit_stress.py > $O/fit_stress.txt 2>&1; tail -4 $O/fit_stress.txt
STRESS_SEED=508 STRESS_TRIALS=200 timeout 900 python tools/eri_stress.py > $O/eri_stress.txt 2>&1; tail -4 $O/eri_stress.txt
STRESS_SEED=509 STRESS_TRIALS=60 timeout 900 python tools/hot_stress.py > $O/hot_stress.txt 2>&1; tail -4 $O/hot_stress.txt
STRESS_SEED=510 STRESS_TRIALS=150 timeout 600 python tools/bcs_stress.py > $O/bcs_stress.txt 2>&1; tail -3 $O/bcs_stress.txt
STRESS_SEED=511 STRESS_TRIALS=150 timeout 600 python tools/gso_stress.py > $O/gso_stress.txt 2>&1; tail -3 $O/gso_stress.txt
