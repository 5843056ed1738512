cd $GRAFT_REPO_ROOT
export M3ASR_LIB=$PWD/tools/_diag_gemm.so
for args in "50 1024 512 ln" "50 1024 512 ln cold" "50 512 1024" "50 512 1024 cold" "50 512 512 cold" "50 1536 512 ln cold"; do
  timeout -k 10 120 python tools/diag_gemm_f32.py $args 2>&1 | grep -v amdgpu.ids
done
