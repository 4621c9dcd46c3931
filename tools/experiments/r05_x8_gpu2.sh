for a in "" "fuse_dbg=6" "fuse_dbg=1" "fuse_dbg=2" "fuse_dbg=4" "fuse_x8_cap=8192" "fuse_x8_cap=16384" "fuse_x8_cap=32768"; do
  echo "== $a"; timeout -k 10 120 python tools/run_iqft_exact.py $a
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_x8_probe2.txt
