for a in "" "fuse_dbg=6" "fuse_dbg=1" "fuse_x8=0" "fuse_x8_cap=8192" "fuse_x8_cap=2048" "fuse_x8_cap=512" ""; do
  echo "== $a"; timeout -k 10 120 python tools/run_iqft_exact.py $a
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_x8_probe3.txt
