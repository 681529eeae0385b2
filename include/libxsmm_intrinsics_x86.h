/* libxsmm_amd: the include name some of the reference's sample drivers pull in for their (compiled-out, "#if 0") AVX-512 gold
 * loops [ref: include/libxsmm_intrinsics_x86.h; samples/equation/equation_layernorm.c:12].  There is no x86 code path in this
 * library -- the kernels are gfx950 -- so the header carries only the host-side declaration helper those drivers use outside the
 * compiled-out blocks. */
#ifndef LIBXSMM_AMD_INTRINSICS_X86_H
#define LIBXSMM_AMD_INTRINSICS_X86_H

#include "libxsmm.h"

/* LIBXSMM_ALIGNED(declaration, bytes): an aligned automatic/static object [ref: include/libxsmm_macros.h LIBXSMM_ALIGNED] */
#if !defined(LIBXSMM_ALIGNED)
# define LIBXSMM_ALIGNED(DECL, N) __attribute__((aligned(N))) DECL
#endif

#endif
