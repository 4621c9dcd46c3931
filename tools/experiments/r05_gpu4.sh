python -m pytest tests/test_gpu_fusion.py tests/test_gpu_basis_front.py -x -q -m gpu > gpurun_out/r05_tests4.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05_tests4.log
for a in "" "fuse_x8_ratio=6" "fuse_dbg=1"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids
for a in "" "fuse_dbg=1" "fuse_dbg=6" "fuse_q3_cap=4096" "fuse_chain_dir=1"; do echo "== tol $a"; timeout -k 10 120 python tools/run_iqft_tol.py $a; done 2>&1 | grep -v amdgpu.ids
