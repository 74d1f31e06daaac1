cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
python3 -m pytest tests/test_gpu_c_host.py -q -m gpu 2>&1 | tail -15
gcc -O2 -Iinclude examples/c_host_eri.c -Llibdmet_preview_amd -l:libdmetk.so -Wl,-rpath,$PWD/libdmet_preview_amd -lm -o /tmp/c_host_eri && /tmp/c_host_eri | tee $O/c_host_eri.txt
