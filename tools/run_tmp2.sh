O=gpurun_out/r04_22; mkdir -p $O
MDH_LIB=$PWD/metdetpy_amd/libmetdet_hip_ps256.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide or format or m3" > $O/pytest_ps256.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_ps256.log | tail -3
bash tools/run_tmp.sh r04_22 "def=metdetpy_amd/libmetdet_hip.so ps16=metdetpy_amd/libmetdet_hip_ps16.so ps64=metdetpy_amd/libmetdet_hip_ps64.so ps256=metdetpy_amd/libmetdet_hip_ps256.so ps1024=metdetpy_amd/libmetdet_hip_ps1024.so" "C2"
