// gemm_wgp_w8_kernels.hip -- 8-bit weights x bf16 activations on the workgroup-per-problem kernel (gemm_wgp.hpp, AK = 0..4).
// Compiled SIX times (Makefile): once plain -- the shape check and the dispatch over the weight kind, no kernels -- and once per kind with -DWGP_W8_KIND=k, which emits
// that kind's seven kernel instances and its launch function.  As one unit the 35 instances took 290 s of a fresh build's critical path; the slowest part now takes 60 s.
#include "gemm_wgp.hpp"

namespace xamd {

// one weight kind: tiles per wave / deal -> instance
template <int KIND>
int launch_wgp_w8_kind(const GemmArgs& a, const Wgp16Geo& g, unsigned int lds_bytes, int tpw, int deal, dim3 grid, dim3 block, hipStream_t st);

#if defined(WGP_W8_KIND)
template <>
int launch_wgp_w8_kind<WGP_W8_KIND>(const GemmArgs& a, const Wgp16Geo& g, unsigned int lds_bytes, int tpw, int deal, dim3 grid, dim3 block, hipStream_t st) {
#define WGPW_(T_, D_) hipLaunchKernelGGL((gemm_wgp16_kernel<false, T_, WGP_W8_KIND, D_>), grid, block, lds_bytes, st, a, g)
#define WGPWD_(T_) do { if (deal == 1) WGPW_(T_, 1); else if (deal == 2) WGPW_(T_, 2); else WGPW_(T_, 0); } while (0)
  if (tpw == 4) WGPW_(4, 1); else if (tpw == 1) WGPW_(1, 0); else if (tpw == 2) WGPWD_(2); else WGPWD_(3);
#undef WGPWD_
#undef WGPW_
  return (int)hipGetLastError();
}
#else
template <> int launch_wgp_w8_kind<0>(const GemmArgs&, const Wgp16Geo&, unsigned int, int, int, dim3, dim3, hipStream_t);
template <> int launch_wgp_w8_kind<1>(const GemmArgs&, const Wgp16Geo&, unsigned int, int, int, dim3, dim3, hipStream_t);
template <> int launch_wgp_w8_kind<2>(const GemmArgs&, const Wgp16Geo&, unsigned int, int, int, dim3, dim3, hipStream_t);
template <> int launch_wgp_w8_kind<3>(const GemmArgs&, const Wgp16Geo&, unsigned int, int, int, dim3, dim3, hipStream_t);
template <> int launch_wgp_w8_kind<4>(const GemmArgs&, const Wgp16Geo&, unsigned int, int, int, dim3, dim3, hipStream_t);

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
  switch (kind) {
    case 0: return launch_wgp_w8_kind<0>(a, g, lds_bytes, tpw, deal, grid, block, st);
    case 1: return launch_wgp_w8_kind<1>(a, g, lds_bytes, tpw, deal, grid, block, st);
    case 2: return launch_wgp_w8_kind<2>(a, g, lds_bytes, tpw, deal, grid, block, st);
    case 3: return launch_wgp_w8_kind<3>(a, g, lds_bytes, tpw, deal, grid, block, st);
    default: return launch_wgp_w8_kind<4>(a, g, lds_bytes, tpw, deal, grid, block, st);
  }
}
#endif

}  // namespace xamd
