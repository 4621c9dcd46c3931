# round 5: does the 8-waves-per-SIMD build of k_fused_rounds (16 walk SGPRs beyond the allocator's budget: 96 SGPRs = 7 blocks of 256
# threads per CU by the guide's residency rule) beat the 7-wave build, whose s72-s87 lie inside the budget?
for a in "fuse_x8=0 fuse_rounds_occ=8" "fuse_x8=0 fuse_rounds_occ=7" "fuse_x8=0 fuse_rounds_occ=6"; do echo "== iqft28 exact, radix-4 walk: $a"; timeout -k 10 120 python tools/run_iqft_exact.py $a; done 2>&1 | grep -v amdgpu.ids
for a in "fuse_rounds_occ=8" "fuse_rounds_occ=7"; do echo "== shor30: $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids
