set -o pipefail
python -m pytest tests/test_gpu_fusion.py -x -q -m gpu > gpurun_out/r05_x8_fusion_tests.log 2>&1; echo "fusion tests rc=$?"; tail -3 gpurun_out/r05_x8_fusion_tests.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config3" > gpurun_out/r05_x8_config3_tests.log 2>&1; echo "config3 tests rc=$?"; tail -3 gpurun_out/r05_x8_config3_tests.log
for a in "" "fuse_x8=0" "fuse_x8_map=0" "fuse_x8_T=11 fuse_x8_c=4" "fuse_chain_dir=0" "fuse_chain_dir=1" "fuse_dbg=1" "fuse_dbg=1 fuse_chain_dir=0"; do
  echo "== $a"; timeout -k 10 120 python tools/run_iqft_exact.py $a
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_x8_probe.txt
