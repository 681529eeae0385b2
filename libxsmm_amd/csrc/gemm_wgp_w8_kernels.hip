// gemm_wgp_w8_kernels.hip -- 8-bit weights x bf16 activations on the workgroup-per-problem kernel (gemm_wgp.hpp, AK = 0..4): a translation unit of its own (30 instances)
#include "gemm_wgp.hpp"

namespace xamd {

// 8-bit weights x bf16 activations on ragged / several-tile shapes (kind as in launch_gemm's P_W8 case); plain strided batches, one block per problem or STRIDE chains
int launch_gemm_wgp16_w8(const GemmArgs& a_in, int kind, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  Wgp16Geo g; unsigned int lds_bytes = 0; int tpw = 0;
  if (kind < 0 || kind > 4 || a_in.b_type != LIBXSMM_DATATYPE_BF16) return 0;
  if (!wgp16_shape_ok(a_in, g, lds_bytes, tpw, kind)) return 0;
  if (kind == 4 && (!a_in.a_scf || (a_in.bs_scf & 3))) return 0;
  GemmArgs a = a_in;
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32; a.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  const int deal = wgp_deal(a.tiles_m, a.tiles_n, tpw);
  const dim3 grid(a.nbatch), block(64u * wgp_waves(a.tiles_m, a.tiles_n, deal));
  *taken = 1;
  if (kernel_name) *kernel_name = "gemm_w8_wgp_kernel";
#define WGPW_(K_, T_, D_) hipLaunchKernelGGL((gemm_wgp16_kernel<false, T_, K_, D_>), grid, block, lds_bytes, st, a, g)
#define WGPWD_(K_, T_) do { if (deal == 1) WGPW_(K_, T_, 1); else if (deal == 2) WGPW_(K_, T_, 2); else WGPW_(K_, T_, 0); } while (0)
#define WGPWT_(K_) do { if (tpw == 1) WGPW_(K_, 1, 0); else if (tpw == 2) WGPWD_(K_, 2); else WGPWD_(K_, 3); } while (0)
  switch (kind) { case 0: WGPWT_(0); break; case 1: WGPWT_(1); break; case 2: WGPWT_(2); break; case 3: WGPWT_(3); break; default: WGPWT_(4); break; }
#undef WGPWT_
#undef WGPWD_
#undef WGPW_
  return (int)hipGetLastError();
}

}  // namespace xamd
