cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_fuzz.py -x -q -k "windows_8_to_31 or fuzz or other_threshold or strided" > gpurun_out/job34_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/job34_tests.log
timeout -k 10 300 python tools/window_probe.py 256 7 8 15 16 17 20 24 28 31 32 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/job34_windows.txt
