cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for tile in 128 256; do for lib in libopenglue_amd libog_gabl_1; do echo "tile=$tile lib=$lib"; OG_GEMM_TILE=$tile OPENGLUE_AMD_LIB=$PWD/openglue_amd/lib/$lib.so timeout 200 python scripts/bench_gemm_scaling.py | grep -E "tiles=  256|tiles=  512|tiles=    8"; done; done
