/*
 * ref_shim.c -- builds the REAL reference (libxsmm/libxsmm under /root/reference) into
 * oracle/_ref/libxsmm_ref.so so that tests can (1) pin the restatement in oracle_*.c against
 * it and (2) time the reference's own CPU JIT path as bench.py's cpu_baseline("reference").
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into libxsmm_amd.  No reference source is copied:
 * this translation unit #includes the reference's header-only entry point from where it lies
 * (the include path is given by oracle/Makefile) and re-exports a handful of its functions
 * under an `xref_` prefix so both libraries can live in one process.
 *
 * Struct layouts/enums are the reference's own here (its headers are in scope), which is also
 * what makes this shim a layout check for include/libxsmm.h: tests compare sizeof()s.
 */
#include <libxsmm_source.h>
#include <stddef.h>

#define XREF __attribute__((visibility("default")))

XREF void xref_init(void) { libxsmm_init(); }
XREF void xref_finalize(void) { libxsmm_finalize(); }
XREF const char* xref_get_target_arch(void) { return libxsmm_get_target_arch(); }
XREF void xref_set_target_arch(const char* arch) { libxsmm_set_target_arch(arch); }

/* layout probe: sizes the GPU library's header must reproduce */
XREF void xref_struct_sizes(size_t* out, int n) {
  const size_t sizes[] = {
    sizeof(libxsmm_gemm_param), sizeof(libxsmm_gemm_ext_param), sizeof(libxsmm_matrix_arg), sizeof(libxsmm_matrix_op_arg),
    sizeof(libxsmm_meltw_unary_param), sizeof(libxsmm_meltw_binary_param), sizeof(libxsmm_meltw_ternary_param),
    sizeof(libxsmm_gemm_shape), sizeof(libxsmm_gemm_batch_reduce_config), sizeof(libxsmm_gemm_ext_unary_argops),
    sizeof(libxsmm_gemm_ext_binary_postops), sizeof(libxsmm_meltw_unary_shape), sizeof(libxsmm_meltw_binary_shape),
    sizeof(libxsmm_meltw_ternary_shape), sizeof(libxsmm_spgemm_config), sizeof(libxsmm_kernel_info),
    sizeof(libxsmm_mmkernel_info), sizeof(libxsmm_descriptor_blob),
    sizeof(libxsmm_matdiff_info), offsetof(libxsmm_matdiff_info, rsq), offsetof(libxsmm_matdiff_info, v_ref), offsetof(libxsmm_matdiff_info, m),
    sizeof(libxsmm_meqn_param), sizeof(libxsmm_meqn_arg_shape), sizeof(libxsmm_matrix_arg_attributes), sizeof(libxsmm_meqn_op_metadata),
    sizeof(libxsmm_meltwkernel_info), sizeof(libxsmm_registry_info), offsetof(libxsmm_gemm_ext_param, d), offsetof(libxsmm_meqn_param, output)
  };
  int i; for (i = 0; i < n && i < (int)(sizeof(sizes) / sizeof(*sizes)); ++i) out[i] = sizes[i];
}

/* --- the reference's C reference implementations (the parity oracle proper) ------------- */
XREF int xref_reference_gemm(const void* param, libxsmm_gemm_shape shape, libxsmm_bitfield flags,
  libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config brcfg)
{
  libxsmm_descriptor_blob blob;
  const libxsmm_gemm_descriptor* desc = libxsmm_gemm_descriptor_init_brgemm(&blob, shape, flags, prefetch, brcfg);
  if (NULL == desc) return -1;
  libxsmm_reference_gemm((void*)param, desc);
  return 0;
}
XREF int xref_reference_gemm_ext(const void* param, libxsmm_gemm_shape shape, libxsmm_bitfield flags,
  libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config brcfg,
  libxsmm_gemm_ext_unary_argops argops, libxsmm_gemm_ext_binary_postops postops)
{
  libxsmm_descriptor_blob blob;
  const libxsmm_gemm_descriptor* desc = libxsmm_gemm_descriptor_init_brgemm_ext(&blob, shape, flags, prefetch, brcfg, argops, postops);
  if (NULL == desc) return -1;
  libxsmm_reference_gemm((void*)param, desc);
  return 0;
}
XREF int xref_reference_meltw_unary(const void* param, libxsmm_meltw_unary_type type, libxsmm_meltw_unary_shape s, libxsmm_bitfield flags) {
  libxsmm_descriptor_blob blob;
  const libxsmm_meltw_descriptor* desc = libxsmm_meltw_descriptor_init2(&blob, s.in0_type, LIBXSMM_DATATYPE_UNSUPPORTED,
    LIBXSMM_DATATYPE_UNSUPPORTED, s.comp_type, s.out_type, s.m, s.n, s.ldi, s.ldo, 0, 0,
    (unsigned short)flags, (unsigned short)type, LIBXSMM_MELTW_OPERATION_UNARY);
  libxsmm_reference_elementwise((void*)param, desc);
  return 0;
}
/* rows per draw of the DROPOUT generator = the reference's 32-bit vector length on this CPU [ref: mateltwise ref :2369] */
XREF int xref_vlen32(void) { return libxsmm_cpuid_vlen32(libxsmm_get_target_archid()); }
XREF int xref_reference_meltw_binary(const void* param, libxsmm_meltw_binary_type type, libxsmm_meltw_binary_shape s, libxsmm_bitfield flags) {
  libxsmm_descriptor_blob blob;
  const libxsmm_meltw_descriptor* desc = libxsmm_meltw_descriptor_init2(&blob, s.in0_type, s.in1_type,
    LIBXSMM_DATATYPE_UNSUPPORTED, s.comp_type, s.out_type, s.m, s.n, s.ldi, s.ldo, s.ldi2, 0,
    (unsigned short)flags, (unsigned short)type, LIBXSMM_MELTW_OPERATION_BINARY);
  libxsmm_reference_elementwise((void*)param, desc);
  return 0;
}
XREF int xref_reference_meltw_ternary(const void* param, libxsmm_meltw_ternary_type type, libxsmm_meltw_ternary_shape s, libxsmm_bitfield flags) {
  libxsmm_descriptor_blob blob;
  const libxsmm_meltw_descriptor* desc = libxsmm_meltw_descriptor_init2(&blob, s.in0_type, s.in1_type, s.in2_type,
    s.comp_type, s.out_type, s.m, s.n, s.ldi, s.ldo, s.ldi2, s.ldi3,
    (unsigned short)flags, (unsigned short)type, LIBXSMM_MELTW_OPERATION_TERNARY);
  libxsmm_reference_elementwise((void*)param, desc);
  return 0;
}

/* --- the reference's CPU JIT dispatch (CPU baseline + second opinion) ----------------------- */
XREF libxsmm_gemmfunction xref_dispatch_gemm(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch) {
  return libxsmm_dispatch_gemm(shape, flags, prefetch);
}
XREF libxsmm_gemmfunction xref_dispatch_brgemm(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_gemm_batch_reduce_config brcfg) {
  return libxsmm_dispatch_brgemm(shape, flags, prefetch, brcfg);
}
XREF libxsmm_gemmfunction_ext xref_dispatch_brgemm_ext(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_gemm_batch_reduce_config brcfg, libxsmm_gemm_ext_unary_argops argops, libxsmm_gemm_ext_binary_postops postops) {
  return libxsmm_dispatch_brgemm_ext(shape, flags, prefetch, brcfg, argops, postops);
}
XREF libxsmm_meltwfunction_unary xref_dispatch_meltw_unary(libxsmm_meltw_unary_type t, libxsmm_meltw_unary_shape s, libxsmm_bitfield f) {
  return libxsmm_dispatch_meltw_unary(t, s, f);
}
XREF libxsmm_meltwfunction_binary xref_dispatch_meltw_binary(libxsmm_meltw_binary_type t, libxsmm_meltw_binary_shape s, libxsmm_bitfield f) {
  return libxsmm_dispatch_meltw_binary(t, s, f);
}
XREF libxsmm_meltwfunction_ternary xref_dispatch_meltw_ternary(libxsmm_meltw_ternary_type t, libxsmm_meltw_ternary_shape s, libxsmm_bitfield f) {
  return libxsmm_dispatch_meltw_ternary(t, s, f);
}
XREF libxsmm_gemmfunction xref_create_packed_spgemm_csr(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* row_ptr, const unsigned int* column_idx, const void* values) {
  return libxsmm_create_packed_spgemm_csr(shape, flags, prefetch, packed_width, row_ptr, column_idx, values);
}
XREF libxsmm_gemmfunction xref_create_packed_spgemm_csc(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_blasint packed_width, const unsigned int* column_ptr, const unsigned int* row_idx, const void* values) {
  return libxsmm_create_packed_spgemm_csc(shape, flags, prefetch, packed_width, column_ptr, row_idx, values);
}
XREF libxsmm_gemmfunction xref_create_packed_spgemm_bcsc(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch,
  libxsmm_spgemm_config cfg) {
  return libxsmm_create_packed_spgemm_bcsc(shape, flags, prefetch, cfg);
}
XREF void xref_release_kernel(const void* kernel) { libxsmm_release_kernel(kernel); }
XREF int xref_get_kernel_info(const void* kernel, libxsmm_kernel_info* info) { return libxsmm_get_kernel_info(kernel, info); }

XREF libxsmm_fsspmdm* xref_fsspmdm_create(libxsmm_datatype datatype, libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, const void* a_dense,
  int c_is_nt, libxsmm_timer_tickint (*timer_tick)(void)) {
  return libxsmm_fsspmdm_create(datatype, M, N, K, lda, ldb, ldc, alpha, beta, a_dense, c_is_nt, timer_tick);
}
XREF void xref_fsspmdm_execute(const libxsmm_fsspmdm* handle, const void* B, void* C) { libxsmm_fsspmdm_execute(handle, B, C); }
XREF void xref_fsspmdm_destroy(libxsmm_fsspmdm* handle) { libxsmm_fsspmdm_destroy(handle); }

/* --- small helpers the pin tests use ------------------------------------------------------------ */
XREF unsigned short xref_convert_f32_to_bf16_rne(float x) { return libxsmm_convert_f32_to_bf16_rne(x); }
XREF unsigned short xref_convert_f32_to_bf16_truncate(float x) { return libxsmm_convert_f32_to_bf16_truncate(x); }
XREF float xref_convert_bf16_to_f32(unsigned short x) { return libxsmm_convert_bf16_to_f32(x); }
/* 16/8-bit float helpers of include/libxsmm_utils.h, pinned in tests/test_utils_cpu.py */
XREF unsigned short xref_convert_f32_to_f16(float x) { return libxsmm_convert_f32_to_f16(x); }
XREF float xref_convert_f16_to_f32(unsigned short x) { return libxsmm_convert_f16_to_f32(x); }
XREF unsigned char xref_convert_f32_to_bf8_rne(float x) { return libxsmm_convert_f32_to_bf8_rne(x); }
XREF unsigned char xref_convert_f16_to_hf8_rne(unsigned short x) { return libxsmm_convert_f16_to_hf8_rne(x); }
XREF unsigned char xref_convert_f32_to_hf8_rne(float x) { return libxsmm_convert_f32_to_hf8_rne(x); }
XREF unsigned char xref_convert_f32_to_bf8_stochastic(float x, unsigned int seed) { return libxsmm_convert_f32_to_bf8_stochastic(x, seed); }
XREF float xref_convert_bf8_to_f32(unsigned char x) { return libxsmm_convert_bf8_to_f32(x); }
XREF float xref_convert_hf8_to_f32(unsigned char x) { return libxsmm_convert_hf8_to_f32(x); }
XREF int xref_cpuid_dot_pack_factor(libxsmm_datatype t) { return libxsmm_cpuid_dot_pack_factor(t); }
/* the seeded generators the drivers build their data with (tests/test_utils_cpu.py) */
XREF void xref_rng_set_seed(unsigned int seed) { libxsmm_rng_set_seed(seed); }
XREF double xref_rng_f64(void) { return libxsmm_rng_f64(); }
XREF unsigned int xref_rng_u32(unsigned int n) { return libxsmm_rng_u32(n); }
XREF void xref_rng_seq(void* data, size_t nbytes) { libxsmm_rng_seq(data, nbytes); }
XREF void xref_rng_f32_seq(float* r, int count) { libxsmm_rng_f32_seq(r, count); }
XREF unsigned int* xref_rng_create_extstate(unsigned int seed) { return libxsmm_rng_create_extstate(seed); }
XREF void xref_rng_destroy_extstate(unsigned int* st) { libxsmm_rng_destroy_extstate(st); }
XREF void xref_stochastic_convert_fp32_bf8(const float* in, unsigned char* out, unsigned int n, void* st, unsigned int start) { libxsmm_stochastic_convert_fp32_bf8(in, out, n, st, start); }
/* the whole statistics record, for tests/test_utils_cpu.py (same struct layout on both sides, checked by test_capi_cpu.py) */
XREF int xref_matdiff(void* info, libxsmm_datatype t, libxsmm_blasint m, libxsmm_blasint n, const void* ref, const void* tst, const libxsmm_blasint* ldref, const libxsmm_blasint* ldtst) {
  return libxsmm_matdiff((libxsmm_matdiff_info*)info, t, m, n, ref, tst, ldref, ldtst);
}
XREF void xref_matdiff_reduce(void* out, const void* in) { libxsmm_matdiff_reduce((libxsmm_matdiff_info*)out, (const libxsmm_matdiff_info*)in); }
XREF void xref_matdiff_clear(void* info) { libxsmm_matdiff_clear((libxsmm_matdiff_info*)info); }
XREF double xref_matdiff_epsilon(const void* info) { return libxsmm_matdiff_epsilon((const libxsmm_matdiff_info*)info); }
XREF double xref_matdiff_normf_rel(libxsmm_datatype t, libxsmm_blasint m, libxsmm_blasint n, const void* ref, const void* tst) {
  libxsmm_matdiff_info info; libxsmm_matdiff_clear(&info);
  if (EXIT_SUCCESS != libxsmm_matdiff(&info, t, m, n, ref, tst, NULL, NULL)) return -1.0;
  return info.normf_rel;
}

/* Time `reps` back-to-back calls of a JIT'ed (BR)GEMM over `count` independent problems laid out
 * with byte strides -- the caller's loop of the reference (documentation/libxsmm_mm.md:95-107),
 * single-threaded.  Returns seconds.  Used only by bench.py's cpu_baseline leg. */
XREF double xref_time_gemm_batch(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param, size_t count,
  long long sa, long long sb, long long sc, int reps)
{
  libxsmm_timer_tickint t0, t1; int r; size_t i;
  libxsmm_gemm_param q = *param;
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps; ++r) {
    for (i = 0; i < count; ++i) {
      q.a.primary = (char*)param->a.primary + (long long)i * sa;
      q.b.primary = (char*)param->b.primary + (long long)i * sb;
      q.c.primary = (char*)param->c.primary + (long long)i * sc;
      kernel(&q);
    }
  }
  t1 = libxsmm_timer_tick();
  return libxsmm_timer_duration(t0, t1);
}

XREF libxsmm_gemmfunction xref_create_packed_gemm(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  return libxsmm_create_packed_gemm(shape, flags, prefetch, packed_width);
}
XREF libxsmm_gemmfunction xref_create_packed_gemm_ac_rm(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  return libxsmm_create_packed_gemm_ac_rm(shape, flags, prefetch, packed_width);
}
XREF libxsmm_gemmfunction xref_create_packed_gemm_bc_rm(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_blasint packed_width) {
  return libxsmm_create_packed_gemm_bc_rm(shape, flags, prefetch, packed_width);
}

/* Timing helpers for the CPU-baseline legs of tools/bench_paths.py (single thread, back-to-back calls). */
XREF double xref_time_gemm_ext_batch(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* param, size_t count,
  long long sa, long long sb, long long sc, int reps)
{
  libxsmm_timer_tickint t0, t1; int r; size_t i;
  libxsmm_gemm_ext_param q = *param;
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps; ++r) {
    for (i = 0; i < count; ++i) {
      q.a.primary = (char*)param->a.primary + (long long)i * sa;
      q.b.primary = (char*)param->b.primary + (long long)i * sb;
      q.c.primary = (char*)param->c.primary + (long long)i * sc;
      kernel(&q);
    }
  }
  t1 = libxsmm_timer_tick();
  return libxsmm_timer_duration(t0, t1);
}
XREF double xref_time_fsspmdm(const libxsmm_fsspmdm* handle, const void* B, void* C, int reps)
{
  libxsmm_timer_tickint t0, t1; int r;
  t0 = libxsmm_timer_tick();
  for (r = 0; r < reps; ++r) libxsmm_fsspmdm_execute(handle, B, C);
  t1 = libxsmm_timer_tick();
  return libxsmm_timer_duration(t0, t1);
}

/* matrix equations [ref: include/libxsmm.h:149-162] */
XREF libxsmm_blasint xref_meqn_create(void) { return libxsmm_meqn_create(); }
XREF libxsmm_meqn_arg_shape xref_create_meqn_arg_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ld, libxsmm_datatype type) { return libxsmm_create_meqn_arg_shape(m, n, ld, type); }
XREF libxsmm_matrix_arg_attributes xref_create_matrix_arg_attributes(libxsmm_matrix_arg_type type, libxsmm_matrix_arg_set_type set_type, libxsmm_blasint card, libxsmm_blasint stride) {
  return libxsmm_create_matrix_arg_attributes(type, set_type, card, stride);
}
XREF libxsmm_meqn_arg_metadata xref_create_meqn_arg_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint in_arg_pos) { return libxsmm_create_meqn_arg_metadata(eqn_idx, in_arg_pos); }
XREF libxsmm_meqn_op_metadata xref_create_meqn_op_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint op_arg_pos) { return libxsmm_create_meqn_op_metadata(eqn_idx, op_arg_pos); }
XREF int xref_meqn_push_back_arg(libxsmm_meqn_arg_metadata md, libxsmm_meqn_arg_shape shape, libxsmm_matrix_arg_attributes attr) { return libxsmm_meqn_push_back_arg(md, shape, attr); }
XREF int xref_meqn_push_back_unary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_unary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) { return libxsmm_meqn_push_back_unary_op(md, type, dtype, flags); }
XREF int xref_meqn_push_back_binary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_binary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) { return libxsmm_meqn_push_back_binary_op(md, type, dtype, flags); }
XREF int xref_meqn_push_back_ternary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_ternary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) { return libxsmm_meqn_push_back_ternary_op(md, type, dtype, flags); }
XREF libxsmm_meqn_function xref_dispatch_meqn(libxsmm_blasint idx, libxsmm_meqn_arg_shape out_shape) { return libxsmm_dispatch_meqn(idx, out_shape); }

XREF void xref_dgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb, const double* beta, double* c, const libxsmm_blasint* ldc) {
  libxsmm_dgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
XREF void xref_sgemm(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb, const float* beta, float* c, const libxsmm_blasint* ldc) {
  libxsmm_sgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}

/* --- differential accept / refuse parity (tests/test_dispatch_differential_cpu.py): what the reference's own descriptor initialisers and dispatcher say about a
 * descriptor, and what its introspection reports for the handle [ref: src/libxsmm_generator.c:36-321, src/libxsmm_main.c:3004-3131] ------------------------------- */
XREF int xref_gemm_descriptor_ok(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config brcfg) {
  libxsmm_descriptor_blob blob;
  return NULL != libxsmm_gemm_descriptor_init_brgemm(&blob, shape, flags, prefetch, brcfg);
}
XREF int xref_gemm_ext_descriptor_ok(libxsmm_gemm_shape shape, libxsmm_bitfield flags, libxsmm_bitfield prefetch, libxsmm_gemm_batch_reduce_config brcfg,
  libxsmm_gemm_ext_unary_argops argops, libxsmm_gemm_ext_binary_postops postops) {
  libxsmm_descriptor_blob blob;
  return NULL != libxsmm_gemm_descriptor_init_brgemm_ext(&blob, shape, flags, prefetch, brcfg, argops, postops);
}
XREF int xref_get_mmkernel_info(const void* kernel, libxsmm_mmkernel_info* info) { libxsmm_xmmfunction f; f.ptr_const = kernel; return libxsmm_get_mmkernel_info(f, info); }
XREF int xref_get_meltwkernel_info(const void* kernel, libxsmm_meltwkernel_info* info) { libxsmm_xmeltwfunction f; memset(&f, 0, sizeof(f)); memcpy(&f, &kernel, sizeof(kernel)); return libxsmm_get_meltwkernel_info(f, info); }
XREF libxsmm_tilecfgfunction xref_dispatch_tilecfg_gemm(libxsmm_gemm_shape shape, libxsmm_bitfield flags) { return libxsmm_dispatch_tilecfg_gemm(shape, flags); }
