python -m pytest tests/test_gpu_tolerance_mode.py tests/test_gpu_fusion.py -x -q -m gpu > gpurun_out/r05_tests5.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_tests5.log
for a in "" "fuse_x8t=0" "fuse_dbg=1" "fuse_dbg=6" "fuse_x8_map=0"; do echo "== tol $a"; timeout -k 10 120 python tools/run_iqft_tol.py $a; done 2>&1 | grep -v amdgpu.ids
for a in "" "fuse_x8t=0"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids
