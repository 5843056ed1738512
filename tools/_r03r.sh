cd $GRAFT_REPO_ROOT
for args in "50 512 512 cold" "50 1024 512 ln cold" "50 1536 512 ln cold"; do
  M3ASR_LIB=$PWD/tools/_diag_gemm.so timeout -k 10 120 python tools/diag_gemm_f32.py $args 2>&1 | grep -v amdgpu.ids | grep "work-groups\|placement\|MFMA"
done
