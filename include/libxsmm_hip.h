/*
 * libxsmm_hip.h -- the GPU-side additions to the LIBXSMM dispatch/param API.
 *
 * Nothing in here exists in the reference: the reference executes one tiny kernel per
 * synchronous call on host pointers and leaves batching/threading to the caller's
 * OpenMP loop [ref: documentation/libxsmm_mm.md:95-107].  A GPU needs the caller's
 * loop *inside* one launch, a stream to order work on, and device memory.  These entry
 * points provide exactly that and nothing else; every one of them is plain C ABI
 * (pointers and sizes, no HIP or torch types in the signatures).
 *
 * Semantics of the batched launchers -- the contract the parity tests check:
 *
 *   libxsmm_hip_gemm_batch_strided(f, p, count, sa, sb, sc)
 *     ==  for (i = 0; i < count; ++i) { q = *p;
 *            q.a.primary = (char*)p->a.primary + i*sa;
 *            q.b.primary = (char*)p->b.primary + i*sb;
 *            q.c.primary = (char*)p->c.primary + i*sc;  f(&q); }
 *
 * i.e. the caller's loop over independent (BR)GEMMs, with byte strides applied to the
 * `primary` slots (a stride of 0 shares the operand across the batch, e.g. weights).
 * In BATCH_REDUCE_ADDRESS mode a/b.primary are pointer arrays, so sa/sb step through
 * those arrays (sa = br_count*sizeof(void*) gives each batch element its own list).
 * op.tertiary (br_count), a/b.secondary (offset arrays) are shared by all elements.
 * MXFP4 weights: the E8M0 scales in a.tertiary step with A -- by sa*2/32 bytes (one scale byte per 32 weights; sa must be
 * a multiple of 16), or by sa through the list of per-block scale pointers in BATCH_REDUCE_ADDRESS mode.
 *
 * The same call accepts a packed sparse handle (libxsmm_create_packed_spgemm_csr/_csc, _spgemm_csr_areg, FsSpMDM kernels):
 * the loop over element-local packed tensors an application like EDGE runs around one small operator
 * [ref: samples/edge]. The sparse operand's values are shared, so its stride must be 0 (anything else is an error);
 * the dense operand and C step by sb/sa and sc.  A kernel whose single call was too small to specialise is
 * specialised by the first eager batched launch that covers enough columns.
 */
#ifndef LIBXSMM_HIP_H
#define LIBXSMM_HIP_H

#if !defined(LIBXSMM_H)
# error include libxsmm.h, not libxsmm_hip.h
#endif

/* ---- device, stream and synchronisation policy (state is per host thread) --------- */
/** Number of visible HIP devices (0 if none: every dispatch then returns NULL). */
LIBXSMM_API int libxsmm_hip_device_count(void);
/** Select the device used by the calling thread for subsequent dispatch/launches. */
LIBXSMM_API int libxsmm_hip_set_device(int device);
LIBXSMM_API int libxsmm_hip_get_device(void);
/**
 * Launch on the given hipStream_t (passed as void*; NULL = the legacy default stream) and
 * switch the calling thread to stream-ordered (asynchronous) execution: a kernel call
 * returns after enqueueing, results are valid in stream order.
 */
LIBXSMM_API void libxsmm_hip_set_stream(void* hip_stream);
LIBXSMM_API void* libxsmm_hip_get_stream(void);
/**
 * 0 (default, also LIBXSMM_HIP_SYNC=1): every kernel call blocks until C is valid, which
 * is the reference's semantics.  1: stream-ordered.  LIBXSMM_HIP_ASYNC=1 presets it.
 * 2 (LIBXSMM_HIP_ASYNC=2 or LIBXSMM_HIP_COALESCE=1): stream-ordered AND coalescing -- the reference leaves the batch loop to the caller
 * (one small GEMM per call [ref: documentation/libxsmm_mm.md:95-107]); in this mode consecutive calls through ONE plain GEMM / stride-BRGEMM handle
 * are queued (three pointers per call, nothing is launched) and leave as ONE pointer-list batch launch when anything else happens: a call through
 * another handle or kind, another batch-reduce count, libxsmm_hip_sync / _set_stream / _set_async / libxsmm_finalize / libxsmm_release_kernel, 65 536
 * queued calls, or a call that reads or writes what a queued call writes (or writes what one reads): the caller's dependent sequences keep their
 * order.  An unmodified `for (i...) kernel(&param_i);` loop followed by libxsmm_hip_sync() thus runs as one batched launch.  Operands must be
 * device-accessible (as in mode 1); calls with a fused operator, pointer / offset lists or per-call scale operands are not queued (they launch as in mode 1).
 */
LIBXSMM_API void libxsmm_hip_set_async(int enable);
LIBXSMM_API int libxsmm_hip_get_async(void);
/**
 * What the calling thread's next launches should assume about their dense operands -- the read-side counterpart of the reference's
 * non-temporal-store hint for C [ref: include/libxsmm_typedefs.h LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT]:
 *   0 (default, also LIBXSMM_HIP_STREAMING=0): decide per launch -- operands of a launch that moves more than the 256 MiB Infinity Cache
 *     holds are loaded non-temporally (they cannot be resident); so are the operands of a smaller launch when the thread's recent launches on OTHER
 *     operand sets (the last 32 sets, each forgotten after 96 launches) together with this one exceed the cache -- a caller that walks over more
 *     input than the cache holds re-reads nothing from it either (round 6); a caller that keeps launching on the same resident set stays cacheable,
 *     and so does a launch whose first operand is what one of those recent launches wrote (a hand-over inside a chain: GEMM, then a TPP on its C);
 *   1: operands are re-read by later launches or were just produced on the device (keep them cacheable, never non-temporal);
 *   2: operands are read once from HBM (a pass over a working set far larger than the cache): non-temporal loads at every size.
 * Measured on 4096 f32 32^3 problems: hint 2 is 7 % faster when the operands do come from HBM and 50 % slower when they were
 * resident in the Infinity Cache, which is why it is not a default; since round 6 mode 0 makes the same choice as the right
 * declaration in both cases (one resident set 5.69 / 5.68 / 9.06 us for modes 0 / 1 / 2, twelve rotated sets 9.97 / 10.74 / 9.99 us).
 * libxsmm_hip_streaming_window_verdict(): what mode 0's look at the recent launches said for the calling thread's last launch (1: they exceed the cache).
 */
LIBXSMM_API void libxsmm_hip_set_streaming_hint(int mode);
LIBXSMM_API int libxsmm_hip_get_streaming_hint(void);
LIBXSMM_API int libxsmm_hip_streaming_window_verdict(void);
/** Block until all work enqueued by the calling thread's stream has finished. */
LIBXSMM_API void libxsmm_hip_sync(void);
/** Pipeline section: the calling thread DECLARES that the kernel launches it issues between _begin and _end are mutually independent (no launch reads
 * what another one of the section writes) and may overlap on the device.  A launch of a few thousand small problems is one round of waves: a third of
 * its time is filling and draining the chip (the headline launch sits on the copy floor of its footprint), and back-to-back launches on one
 * stream cannot overlap that.  Inside a section consecutive launches rotate over `lanes` (2..8) internal streams: _begin forks them off the thread's stream
 * (they wait for everything issued to it before), _end joins them back (the thread's stream waits for every lane); everything is stream-ordered, so a
 * section may be captured into a hipGraph (the lanes become parallel branches).  Needs stream-ordered launches (libxsmm_hip_set_stream / _set_async);
 * libxsmm_hip_sync and libxsmm_hip_set_stream close an open section.  Each lane has its own partial-result workspace. */
LIBXSMM_API int libxsmm_hip_pipeline_begin(int lanes);
LIBXSMM_API int libxsmm_hip_pipeline_end(void);
/** Sticky error state of the calling thread (0 = none); kernels have no error channel. */
LIBXSMM_API int libxsmm_hip_get_last_error(void);
LIBXSMM_API const char* libxsmm_hip_get_last_error_string(void);
LIBXSMM_API void libxsmm_hip_clear_last_error(void);

/* ---- device memory for C callers that do not want to include HIP headers ----------- */
LIBXSMM_API void* libxsmm_hip_malloc(size_t nbytes);
LIBXSMM_API void libxsmm_hip_free(void* device_ptr);
LIBXSMM_API int libxsmm_hip_memcpy_h2d(void* device_dst, const void* host_src, size_t nbytes);
LIBXSMM_API int libxsmm_hip_memcpy_d2h(void* host_dst, const void* device_src, size_t nbytes);
LIBXSMM_API int libxsmm_hip_memset(void* device_dst, int value, size_t nbytes);

/* ---- batched launches: the caller's loop moved into one grid ------------------------ */
LIBXSMM_API void libxsmm_hip_gemm_batch_strided(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param,
  size_t count, long long stride_a, long long stride_b, long long stride_c);
/** Adds byte strides for d.primary (fused bias) and c.secondary (ReLU bitmask). */
LIBXSMM_API void libxsmm_hip_gemm_ext_batch_strided(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* param,
  size_t count, long long stride_a, long long stride_b, long long stride_c, long long stride_d, long long stride_mask);
/**
 * 2-D strided batch -- the two nested loops a caller runs to build a blocked GEMM out of (BR)GEMM tiles
 * [ref: the loop nests around the kernel in samples/xgemm/gemm_kernel.c:3186-3226 and the DL drivers built on BRGEMM]:
 *   for (j = 0; j < count_j; ++j) for (i = 0; i < count_i; ++i) { q = *p;
 *     q.a.primary = (char*)p->a.primary + i*stride_a_i;  q.b.primary = (char*)p->b.primary + j*stride_b_j;
 *     q.c.primary = (char*)p->c.primary + i*stride_c_i + j*stride_c_j;  f(&q); }
 * A is re-used by every j and B by every i: the launch deals contiguous bands of j to the eight XCDs so that the re-use
 * happens in their L2s.  The ext form steps d.primary (column bias) with i and c.secondary (ReLU bitmask) with both.
 * MXFP4 / MX scales step with their operand as in the 1-D form.
 */
LIBXSMM_API void libxsmm_hip_gemm_batch_strided_2d(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param, size_t count_i, size_t count_j,
  long long stride_a_i, long long stride_b_j, long long stride_c_i, long long stride_c_j);
LIBXSMM_API void libxsmm_hip_gemm_ext_batch_strided_2d(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* param, size_t count_i, size_t count_j,
  long long stride_a_i, long long stride_b_j, long long stride_c_i, long long stride_c_j, long long stride_d_i, long long stride_mask_i, long long stride_mask_j);
/**
 * Pointer-list batch: element i uses a_list[i], b_list[i], c_list[i] as its `primary`
 * slots.  The three lists themselves must be device-accessible arrays of `count` pointers.
 */
LIBXSMM_API void libxsmm_hip_gemm_batch_pointers(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* param,
  size_t count, const void* const* a_list, const void* const* b_list, void* const* c_list);
LIBXSMM_API void libxsmm_hip_meltw_unary_batch_strided(libxsmm_meltwfunction_unary kernel, const libxsmm_meltw_unary_param* param,
  size_t count, long long stride_in, long long stride_out, long long stride_aux);
LIBXSMM_API void libxsmm_hip_meltw_binary_batch_strided(libxsmm_meltwfunction_binary kernel, const libxsmm_meltw_binary_param* param,
  size_t count, long long stride_in0, long long stride_in1, long long stride_out);
LIBXSMM_API void libxsmm_hip_meltw_ternary_batch_strided(libxsmm_meltwfunction_ternary kernel, const libxsmm_meltw_ternary_param* param,
  size_t count, long long stride_in0, long long stride_in1, long long stride_in2, long long stride_out);

/* ---- multi-GPU: the batch / packed / N axis is split by contiguous blocks -----------
 * One process per GPU; no collective on the data path.  Rank r of `world` owns
 * [begin, end) of a `count`-long axis (first `count % world` ranks get one extra unit),
 * after rounding shard boundaries to `granule` units (e.g. the packed width's lane tile). */
LIBXSMM_API void libxsmm_hip_shard_range(size_t count, size_t granule, int world, int rank, size_t* begin, size_t* end);

/* ---- multi-GPU from ONE process and ONE host thread (C / C++ hosts: no launcher, no Python) ---------------------------------------
 * The reference scales out through the caller's loop over independent problems [ref: samples/xgemm/gemm_kernel.c:4063-4066,
 * samples/xgemm_sparse_Ainregs/pyfr_driver_asp_reg.c:379-393]; here that loop is cut into one contiguous block per device.
 * A shard = what ONE device does: a kernel handle, the param struct whose pointers name operands RESIDENT ON THAT DEVICE (the first
 * problem of the shard), and either count = 0 (one plain call kernel(param): the P / m_blocks / N splits, where every shard has a
 * handle of its own shape) or count > 0 (a strided batch exactly as libxsmm_hip_gemm[_ext]_batch_strided / libxsmm_hip_meltw_*_batch_strided:
 * stride[] = {a, b, c, d, mask} for (BR)GEMM and packed sparse handles, {in, out, aux} / {in0, in1, out} / {in0, in1, in2, out} for TPPs).
 * libxsmm_hip_launch_shards issues every shard on a stream of its own on the shard's device -- shards overlap, also several on one
 * device -- each with private staging scratch and partial-result workspaces, and (gather_bytes > 0) follows the shard's kernel with ONE
 * copy of gather_bytes from gather_src to gather_dst + gather_dst_offset on gather_device: the source device pushes over its own xGMI
 * link (hipMemcpyPeerAsync), so nshards - 1 links feed the root at once where a ring would be bound by one link per hop.
 * Blocking thread (default): returns when every shard and copy has finished.  Stream-ordered thread (libxsmm_hip_set_stream / _set_async):
 * the shards start behind what the thread's stream holds and the thread's stream continues behind them; libxsmm_hip_sync() waits.
 * Dense dispatch handles run on every device; created sparse kernels (pattern arrays, generated code) belong to the device that was
 * current at creation (libxsmm_hip_set_device) -- create one per device.  Operands must be device memory.  Returns EXIT_SUCCESS / EXIT_FAILURE
 * (libxsmm_hip_get_last_error_string says why). */
typedef struct libxsmm_hip_shard {
  int device;                          /* HIP device that holds this shard's operands */
  const void* kernel;                  /* any libxsmm_*function handle */
  const void* param;                   /* the matching param struct; its pointers are the shard's FIRST problem, in `device`'s memory */
  size_t count;                        /* 0: kernel(param) once;  > 0: strided batch of `count` problems */
  long long stride[5];                 /* byte strides of the batch (kind specific, see above) */
  const void* gather_src;              /* optional result gather: after the kernel, gather_bytes from here ... */
  size_t gather_bytes, gather_dst_offset;   /* ... to gather_dst + gather_dst_offset on gather_device */
  size_t gather_rows, gather_src_pitch, gather_dst_pitch;   /* gather_rows > 1 (round 6): a PITCHED gather -- gather_rows rows of gather_bytes each, the source rows
                                          gather_src_pitch bytes apart, the destination rows gather_dst_pitch: a shard's column block of a row-major result (the P
                                          axis of a packed C, the N axis of an FsSpMDM C) lands in place inside the whole result.  0 / 1: one contiguous copy. */
} libxsmm_hip_shard;
LIBXSMM_API int libxsmm_hip_launch_shards(const libxsmm_hip_shard* shards, int nshards, int gather_device, void* gather_dst);
/** The batch axis cut by libxsmm_hip_shard_range(count, 1, nshards, s): shard s owns problems [begin_s, end_s) and runs on devices[s]
 * (NULL: device s % device_count; a device may appear more than once).  shard_params[s] holds the pointers of problem begin_s in that device's
 * memory (every device holds only its own block of A / B / C; a shared operand -- stride 0 -- is replicated by the caller).
 * gather_dst != NULL: C of all shards is assembled at gather_dst + begin_s * stride_c on gather_device. */
LIBXSMM_API int libxsmm_hip_gemm_batch_strided_sharded(libxsmm_gemmfunction kernel, const libxsmm_gemm_param* shard_params, size_t count,
  long long stride_a, long long stride_b, long long stride_c, int nshards, const int* devices, int gather_device, void* gather_dst);
LIBXSMM_API int libxsmm_hip_gemm_ext_batch_strided_sharded(libxsmm_gemmfunction_ext kernel, const libxsmm_gemm_ext_param* shard_params, size_t count,
  long long stride_a, long long stride_b, long long stride_c, long long stride_d, long long stride_mask, int nshards, const int* devices, int gather_device, void* gather_dst);

/* ---- created (sparse) kernels, sharded (round 6) ---------------------------------------------------------------------------------------
 * Created kernels belong to the device they were created on (pattern arrays, generated code), so a C host that splits P / M-blocks / N over several GPUs
 * needs one handle per device.  These calls do that loop: given the creator's own arguments they cut the parallel axis with libxsmm_hip_shard_range
 * (granule = whole lane tiles), create one kernel per non-empty shard ON that shard's device (devices[s], NULL: s % device_count; a device may appear
 * several times -- virtual shards), and keep them together.  Every shard's operands live on its device in the shard's OWN compact layout:
 *   packed CSR / CSC (axis = the packed width P):   B [K][N][P_s], C [M][N][P_s] (CSR, A sparse);  A [M][K][P_s], C [M][N][P_s] (CSC, B sparse)
 *   BCSC (axis = the M-blocks = shape.m, as the creator takes them): A and C of the shard's M-blocks; the block-sparse B is replicated by the caller
 *   FsSpMDM (axis = N):                              B [K][N_s], C [M][N_s]  (leading dimensions = N_s)
 * libxsmm_hip_sharded_launch: shard_params[i] (i = 0 .. shards - 1, non-empty shards in order) holds shard i's operand pointers exactly as the plain kernel
 * takes them (FsSpMDM: b.primary, c.primary); all shards are issued by libxsmm_hip_launch_shards (one stream per shard, overlapping).  gather_dst != NULL
 * assembles C on gather_device: gather_dst_pitch = 0 places the shards' C blocks back to back (a valid packed layout of independent slabs),
 * gather_dst_pitch > 0 is the byte pitch of one row of the WHOLE result (P * elem for packed C, ldc * elem for FsSpMDM) and every shard's columns land in
 * place (BCSC: C of an M-block is contiguous, the pitch is ignored).  Returns EXIT_SUCCESS / EXIT_FAILURE like libxsmm_hip_launch_shards. */
typedef struct libxsmm_hip_sharded_kernel libxsmm_hip_sharded_kernel;
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_csr_sharded(libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags,
  libxsmm_blasint packed_width, const unsigned int* row_ptr, const unsigned int* column_idx, const void* values, int nshards, const int* devices);
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_csc_sharded(libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags,
  libxsmm_blasint packed_width, const unsigned int* column_ptr, const unsigned int* row_idx, const void* values, int nshards, const int* devices);
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_create_packed_spgemm_bcsc_sharded(libxsmm_gemm_shape gemm_shape, libxsmm_bitfield gemm_flags, libxsmm_bitfield prefetch_flags,
  libxsmm_spgemm_config spgemm_config, int nshards, const int* devices);
LIBXSMM_API libxsmm_hip_sharded_kernel* libxsmm_hip_fsspmdm_create_sharded(libxsmm_datatype datatype, libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda,
  const void* alpha, const void* beta, const void* a_dense, int nshards, const int* devices);
/** number of (non-empty) shards; shard i's device, its range [begin, end) of the split axis (P columns, M-blocks, N columns) and its plain handle
 * (a libxsmm_gemmfunction; owned by the set) */
LIBXSMM_API int libxsmm_hip_sharded_count(const libxsmm_hip_sharded_kernel* set);
LIBXSMM_API int libxsmm_hip_sharded_range(const libxsmm_hip_sharded_kernel* set, int shard, int* device, size_t* begin, size_t* end);
LIBXSMM_API libxsmm_gemmfunction libxsmm_hip_sharded_handle(const libxsmm_hip_sharded_kernel* set, int shard);
LIBXSMM_API int libxsmm_hip_sharded_launch(libxsmm_hip_sharded_kernel* set, const libxsmm_gemm_param* shard_params, int gather_device, void* gather_dst, size_t gather_dst_pitch);
LIBXSMM_API void libxsmm_hip_sharded_destroy(libxsmm_hip_sharded_kernel* set);

/* Result gather onto one GPU without a collective library (one process per GPU on one node).  Every rank exports the device buffer that
 * holds its shard (libxsmm_hip_ipc_export: LIBXSMM_HIP_IPC_HANDLE_BYTES opaque bytes, to be handed to the root by whatever means the
 * application has -- MPI, a file, torch.distributed); the root then pulls all shards, each on its own stream: the sources sit behind
 * different xGMI links of the root, so the copies overlap (up to world - 1 links) where a ring all-gather is bound by one link per hop.
 *   handles     : world x LIBXSMM_HIP_IPC_HANDLE_BYTES, entry r = what rank r exported (entry self_rank is ignored)
 *   self_src    : the root's own shard (plain device pointer)
 *   src_offsets : byte offset of the shard behind each exported pointer (NULL: all 0);  dst_offsets / nbytes: placement and size in dst
 * Returns 0 on success; the copies are complete on return.  The exporting ranks must keep their buffers alive until then. */
#define LIBXSMM_HIP_IPC_HANDLE_BYTES 80
LIBXSMM_API int libxsmm_hip_ipc_export(const void* device_ptr, void* handle);
LIBXSMM_API int libxsmm_hip_gather_shards(void* dst, int world, int self_rank, const void* handles, const void* self_src,
  const size_t* src_offsets, const size_t* dst_offsets, const size_t* nbytes);

/* ---- input preparation (the reference keeps these in its samples) ----------------------------------------
 * libxsmm_hip_mtx_read: Matrix-Market coordinate file -> CSR (by_column = 0: ptr over rows, idx = columns) or CSC (by_column = 1), values as
 * F32 or F64, entries in any order, empty rows / columns allowed [ref: samples/xgemm_norm_packed/common_edge_proxy.h:29-320].
 * libxsmm_hip_bcsc_from_dense: dense K x N operand stored as the reference's BCSC driver stores it (B[n*K + k]) -> colptr / rowidx / block values
 * [blk][bn][bk], all-zero blocks dropped [ref: samples/xgemm_sparse/spmm_kernel.c:306-347].
 * Every output array comes from libxsmm_aligned_malloc (device-visible pinned memory when a device is present): pattern arrays can be passed to
 * libxsmm_create_packed_spgemm_* / the BCSC call and value arrays to a.primary / b.primary as they are; release them with libxsmm_free.
 * Return EXIT_SUCCESS or EXIT_FAILURE (unreadable file, inconsistent header, index out of range, block sizes that do not divide). */
LIBXSMM_API int libxsmm_hip_mtx_read(const char* path, int by_column, libxsmm_datatype value_type, unsigned int** ptr, unsigned int** idx, void** values,
  unsigned int* rows, unsigned int* cols, unsigned int* nnz);
LIBXSMM_API int libxsmm_hip_bcsc_from_dense(libxsmm_datatype type, const void* dense, int K, int N, int bk, int bn,
  unsigned int** colptr, unsigned int** rowidx, void** values, unsigned int* nnzb);

/** BCSC kernels take their block pattern with every call (b.secondary = colptr, b.tertiary = rowidx, b.quaternary -> block-column count
 * [ref: samples/xgemm_sparse/spmm_kernel.c:423-456]) and need it inverted (block id per (block column, k-block)).  A pattern in HOST memory -- the
 * reference's convention -- is recognised by content and its inverted image cached with the kernel (no allocation, no lock on a hit).  A pattern in
 * DEVICE memory cannot be compared without a host round trip: by default it is inverted by a small kernel in front of every call; this function
 * lets the caller promise that the two device arrays do not change until the binding is replaced (NULL, NULL unbinds), so the table is built once,
 * stream-ordered on the calling thread's stream, and calls that pass exactly these pointers launch the GEMM kernel alone.  Outside a graph capture the
 * two arrays are also read once (the call then waits for the stream): the number of blocks and the k-blocks in use let the launcher pick the kernels
 * that keep a small B in LDS, as a host-resident pattern does. */
LIBXSMM_API int libxsmm_hip_bcsc_bind_pattern(libxsmm_gemmfunction kernel, const unsigned int* colptr, const unsigned int* rowidx, unsigned long long n_block_columns);

/* ---- run-time specialisation of the fixed-pattern sparse kernels ---------------------
 * libxsmm_create_packed_spgemm_csr/_csc, libxsmm_create_spgemm_csr_areg and libxsmm_fsspmdm_create can compile a
 * kernel with the sparsity pattern unrolled into the instruction stream (hiprtc), as the reference's JIT does
 * [ref: src/generator_packed_spgemm_csr_asparse_avx_avx2_avx512.c:336-470].  mode 0: never (precompiled LDS-staged
 * kernels), 1: when one call is large enough to repay the compile time (default), 2: always.
 * LIBXSMM_HIP_JIT=0|1|2 presets it.  Applies to kernels created afterwards. */
LIBXSMM_API void libxsmm_hip_set_jit(int mode);
LIBXSMM_API int libxsmm_hip_get_jit(void);

/* ---- introspection used by the tests and the bench --------------------------------- */
/** Name of the device kernel a handle launches for single (batch==0) or batched calls. */
LIBXSMM_API const char* libxsmm_hip_kernel_name(const void* kernel, int batched);
/** Number of kernel launches issued by the calling thread since the last reset. */
LIBXSMM_API unsigned long long libxsmm_hip_launch_count(int reset);
/** Diagnostic: what the matrix pipe of THIS chip sustains under its power budget.  Every wave (one per SIMD on every CU, 16 accumulators) issues
 * `iterations` x 16 MFMAs back to back on REGISTER operands taken from `operands` (device memory, 64 KiB of bf16 or f32 values) -- no LDS, no memory
 * traffic -- on the calling thread's stream.  datatype BF16: v_mfma_f32_32x32x16_bf16, F32: v_mfma_f32_32x32x2_f32.  *flop receives the floating-point
 * operations of the launch; time it with events.  The rate depends on the operand VALUES (zeros: 99 % of the 2.5 PF bf16 figure at 2.36 GHz; the reference
 * drivers' value distribution: 73 % at 1.75 GHz, profiles/r03_bf16_macro_ablation.txt): the roof a GEMM on the same data cannot exceed. */
LIBXSMM_API int libxsmm_hip_probe_mfma(libxsmm_datatype datatype, const void* operands, int iterations, double* flop);
/** 1 if the library was built with the gfx950 code object and a device is present. */
LIBXSMM_API int libxsmm_hip_available(void);

#endif /* LIBXSMM_HIP_H */
