#!/bin/bash
# Registers and spills of the deblock body per planes variant (8 wave pairs, P pictures): k_deblock2_planes<0, 8> holds
# all three variants and reports the worst. Which of them spills? Runs where hipcc is (no GPU needed).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
cat > $T/db_only.hip <<EOS
#include "$R/cedarx_h264_encoder_amd/csrc/enc_k_wavefront.hip"
namespace cedar {
__global__ __launch_bounds__(128 * 8) void k_db_luma_only(Params p) { deblock2_body<0, 1, 8>(p, (int)blockIdx.x); }
__global__ __launch_bounds__(128 * 8) void k_db_chroma_only(Params p) { deblock2_body<0, 2, 8>(p, (int)blockIdx.x); }
__global__ __launch_bounds__(128 * 8) void k_db_both_planes(Params p) { deblock2_body<0, 3, 8>(p, (int)blockIdx.x); }
}
EOS
cd $T
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -I$R/include -I$R/cedarx_h264_encoder_amd/csrc -x hip -c db_only.hip -o db.o --save-temps 2>/dev/null
grep -B2 -A14 "\.name:.*k_db_" db_only-hip-amdgcn-amd-amdhsa-gfx950.s | grep "\.name\|vgpr_count\|vgpr_spill_count\|private_segment" | paste - - - - | sed 's/  */ /g'
rm -rf $T
