python -m pytest tests/test_gpu_basis_front.py tests/test_gpu_fusion.py -x -q -m gpu > gpurun_out/r05_tests8.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_tests8.log
python -m pytest tests/test_gpu_configs_fullsize.py -x -q -m gpu -k "whole_final_state" > gpurun_out/r05_tests8b.log 2>&1; echo "whole-state tests rc=$?"; tail -3 gpurun_out/r05_tests8b.log
for a in "" "fuse_x8_gen3=0" "fuse_dbg=1"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids
