// gemm_wgp16_kernels.hip -- the 16-bit (bf16 / f16) launcher of the workgroup-per-problem kernel (gemm_wgp.hpp: the kernel, its geometry and the notes on its form)
#include "gemm_wgp.hpp"

namespace xamd {

int launch_gemm_wgp16(const GemmArgs& a_in, void* stream, const char** kernel_name, int* taken) {
  *taken = 0;
  Wgp16Geo g; unsigned int lds_bytes = 0; int tpw = 0;
  const bool f16 = a_in.a_type == LIBXSMM_DATATYPE_F16;
  if (f16 && a_in.comp_f16) return 0;       // (halves with any beta / fused operators: round 6 -- the start value is rounded to f16, gemm_wgp16_kernel round_start)
  if (!wgp16_shape_ok(a_in, g, lds_bytes, tpw)) return 0;
  GemmArgs a = a_in;
  a.tiles_m = (a.m + 31) / 32; a.tiles_n = (a.n + 31) / 32; a.map2d_shift = 0;
  hipStream_t st = (hipStream_t)stream;
  *taken = 1;
  if (kernel_name) *kernel_name = f16 ? "gemm_f16_wgp_kernel" : "gemm_bf16_wgp_kernel";
  const dim3 grid(a.nbatch);
  const int deal = wgp_deal(a.tiles_m, a.tiles_n, tpw);
  const dim3 block(64u * wgp_waves(a.tiles_m, a.tiles_n, deal));
#define WGP_(F_, T_, D_) hipLaunchKernelGGL((gemm_wgp16_kernel<F_, T_, -1, D_>), grid, block, lds_bytes, st, a, g)
#define WGPD_(F_, T_) do { if (deal == 1) WGP_(F_, T_, 1); else if (deal == 2) WGP_(F_, T_, 2); else WGP_(F_, T_, 0); } while (0)
  if (tpw == 4) { if (f16) WGP_(true, 4, 1); else WGP_(false, 4, 1); }      // (4 x 4 tiles: wgp_deal returns 1)
  else if (f16) { if (tpw == 1) WGP_(true, 1, 0); else if (tpw == 2) WGPD_(true, 2); else WGPD_(true, 3); }
  else { if (tpw == 1) WGP_(false, 1, 0); else if (tpw == 2) WGPD_(false, 2); else WGPD_(false, 3); }
#undef WGPD_
#undef WGP_
  return (int)hipGetLastError();
}

}  // namespace xamd
