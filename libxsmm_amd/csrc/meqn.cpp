// meqn.cpp -- matrix equations (trees of TPPs behind one function handle).
//
// Reference: the tree is built in pre-order by libxsmm_meqn_push_back_* [ref: src/libxsmm_matrixeqn.c:996-1190,1363-1510],
// every op node produces an intermediate whose shape follows libxsmm_meqn_adjust_tmp_sizes [ref: :869-936] and whose
// datatype is the op's dtype [ref: :289-311]; the evaluator runs the TPP of every op node bottom-up with the children's
// shapes/types as inputs [ref: src/generator_matequation_reference_impl.c:105-227].  Here each op node becomes one launch
// of the TPP kernels of meltw_kernels.hip (same MeltwArgs the standalone TPPs use, so every node inherits their parity),
// intermediates live in the calling thread's device workspace, launches are stream ordered and the handle synchronises
// once at the end (unless the thread is in async mode).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "internal.hpp"

namespace xamd {

enum { EQ_NONE = 0, EQ_ARG = 1, EQ_UNARY = 2, EQ_BINARY = 3, EQ_TERNARY = 4 };

struct EqnNode {
  int kind = EQ_NONE;
  int op = 0, dtype = 0; unsigned int flags = 0; int op_arg_pos = -1;   // op nodes
  int in_pos = -1, set = 0, set_type = 0; long long set_stride = 0;     // arg nodes (set: a strided block list feeding a BRGEMM node)
  int child[3] = {-1, -1, -1}, up = -1;
  int m = 0, n = 0, ld = 0, type = 0;                                   // result shape / datatype of this node
};

struct Equation {
  std::vector<EqnNode> nodes;       // nodes[0] is the root
  int cur = 0;                      // node that takes the next push
  bool constructed = false;
  std::map<std::array<int, 4>, const void*> handles;   // dispatched (m, n, ld, type) -> handle
};

struct EqnStep { MeltwArgs args; int src[3]; int node; int alpha_from_op; int dump_from_op; int idx_from_input; int root_side; bool scalar_arg[3];
  const void* gemm; int br_from_op; int out_loc; bool skip_if_in_place; };   // gemm: dispatched (BR)GEMM handle of a MATMUL / BRGEMM node; out_loc: like src, INT32_MIN = the caller's output   // idx_from_input: GATHER reads its indices from inputs[pos].secondary   // root_side: 1 bitmask, 2 UNZIP offset from output.secondary   // src: >=0 input position, < 0: -(slot+1)
struct EqnPlan {
  bool out_scalar = false;          // a 1 x 1 result (a dot product, a full reduction): callers keep it on their stack -> staged like scalar inputs
  size_t out_scalar_bytes = 4;
  JitKernel* fused = nullptr;       // whole tree as ONE generated kernel (element-wise trees), else the step chain below
  std::vector<int> fused_inputs;    // input positions in kernel-argument order
  std::vector<char> fused_scalar;   // ... and whether that argument is a 1 x 1 scalar (may live in host memory)
  std::vector<int> fused_alphas;    // op_arg positions of scalar op arguments, in kernel-argument order
  std::vector<int> fused_dumps;     // op_arg positions of DUMP destinations (device pointers), behind the alphas
  std::vector<EqnStep> steps;
  std::vector<int> slot_of;         // per node: workspace slot (-1: none)
  size_t slot_bytes = 0; int nslots = 0;
  bool has_gemm = false;            // a MATMUL / BRGEMM step: room for the GEMM kernel's partial sums is reserved behind the slots
};

namespace {

std::mutex g_eqn_lock;
std::vector<Equation*> g_eqns;

int arity(int kind) { return kind == EQ_UNARY ? 1 : kind == EQ_BINARY ? 2 : kind == EQ_TERNARY ? 3 : 0; }

bool is_reduce(int t) {
  return t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX ||
         t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MUL || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD;
}

// MATMUL / BRGEMM nodes [ref: src/libxsmm_matrixeqn.c:1421-1441,1468-1488; semantics: samples/equation/equation_matmul.c:37-62]: the eight
// variants of a family come in the order plain, B_TRANS, A_TRANS, A_TRANS_B_TRANS, A_VNNI, A_VNNI_B_TRANS, A_VNNI_TRANS, A_VNNI_TRANS_B_TRANS
struct GemmNode { bool is_gemm, br, ta, tb, vnni; };
GemmNode gemm_node(int kind, int op) {
  int v = -1; bool br = false;
  if (kind == EQ_BINARY) {
    if (op == LIBXSMM_MELTW_TYPE_BINARY_MATMUL) v = 0;
    else if (op >= LIBXSMM_MELTW_TYPE_BINARY_MATMUL_B_TRANS && op <= LIBXSMM_MELTW_TYPE_BINARY_MATMUL_A_VNNI_TRANS_B_TRANS) v = op - LIBXSMM_MELTW_TYPE_BINARY_MATMUL_B_TRANS + 1;
    else if (op >= LIBXSMM_MELTW_TYPE_BINARY_BRGEMM && op <= LIBXSMM_MELTW_TYPE_BINARY_BRGEMM_A_VNNI_TRANS_B_TRANS) { v = op - LIBXSMM_MELTW_TYPE_BINARY_BRGEMM; br = true; }
  } else if (kind == EQ_TERNARY) {
    if (op == LIBXSMM_MELTW_TYPE_TERNARY_MATMUL) v = 0;
    else if (op >= LIBXSMM_MELTW_TYPE_TERNARY_MATMUL_B_TRANS && op <= LIBXSMM_MELTW_TYPE_TERNARY_MATMUL_A_VNNI_TRANS_B_TRANS) v = op - LIBXSMM_MELTW_TYPE_TERNARY_MATMUL_B_TRANS + 1;
    else if (op >= LIBXSMM_MELTW_TYPE_TERNARY_BRGEMM && op <= LIBXSMM_MELTW_TYPE_TERNARY_BRGEMM_A_VNNI_TRANS_B_TRANS) { v = op - LIBXSMM_MELTW_TYPE_TERNARY_BRGEMM; br = true; }
  }
  if (v < 0) return GemmNode{false, false, false, false, false};
  return GemmNode{true, br, ((v >> 1) & 1) != 0, (v & 1) != 0, v >= 4};
}

// add a node under `cur` and advance `cur` the way libxsmm_meqn_trv_head does: an op takes the next push itself, an
// argument hands it to the nearest ancestor that still misses an operand
int push(Equation& e, const EqnNode& proto) {
  if (e.constructed) return 1;
  if (e.nodes.empty()) {
    if (proto.kind == EQ_ARG) return 1;                      // the root must be an op [ref: :1064-1075]
    e.nodes.push_back(proto); e.cur = 0;
    return 0;
  }
  EqnNode& parent = e.nodes[e.cur];
  int slot = -1;
  for (int c = 0; c < arity(parent.kind); ++c) if (parent.child[c] < 0) { slot = c; break; }
  if (slot < 0) return 1;
  const int id = (int)e.nodes.size();
  e.nodes.push_back(proto);
  e.nodes[id].up = e.cur;
  e.nodes[e.cur].child[slot] = id;
  if (proto.kind != EQ_ARG) { e.cur = id; return 0; }
  int at = e.cur;
  for (;;) {
    const EqnNode& nd = e.nodes[at];
    bool full = true;
    for (int c = 0; c < arity(nd.kind); ++c) if (nd.child[c] < 0) full = false;
    if (!full) { e.cur = at; return 0; }
    if (nd.up < 0) { e.constructed = true; e.cur = at; return 0; }
    at = nd.up;
  }
}

Equation* get(int idx) { return (idx >= 0 && idx < (int)g_eqns.size()) ? g_eqns[idx] : nullptr; }

// result shape / type of every node, children first [ref: libxsmm_matrixeqn.c:869-936, :289-311]
bool infer(Equation& e, int id) {
  EqnNode& nd = e.nodes[id];
  if (nd.kind == EQ_ARG) {   // a strided set only makes sense as the A / B operand of a BRGEMM node
    if (nd.set != 0) {
      const EqnNode* up = nd.up >= 0 ? &e.nodes[nd.up] : nullptr;
      const GemmNode g = up ? gemm_node(up->kind, up->op) : GemmNode{false, false, false, false, false};
      if (!g.is_gemm || !g.br || nd.set_type != LIBXSMM_MATRIX_ARG_SET_TYPE_STRIDE_BASE || up->child[2] == id) return false;
    }
    return nd.m > 0 && nd.n > 0 && nd.ld >= nd.m;
  }
  for (int c = 0; c < arity(nd.kind); ++c) if (nd.child[c] < 0 || !infer(e, nd.child[c])) return false;
  const GemmNode g = gemm_node(nd.kind, nd.op);
  if (g.is_gemm) {   // op(A) is m x k, op(B) is k x n; a ternary node accumulates into (and is stored in) its third operand
    const EqnNode& A = e.nodes[nd.child[0]]; const EqnNode& B = e.nodes[nd.child[1]];
    const int m = g.ta ? A.n : A.m, k = g.ta ? A.m : A.n, n = g.tb ? B.m : B.n, kb = g.tb ? B.n : B.m;
    if (k != kb || (g.vnni && g.ta)) return false;
    nd.m = m; nd.n = n; nd.ld = m; nd.type = nd.dtype;
    if (nd.kind == EQ_TERNARY) {
      const EqnNode& Cn = e.nodes[nd.child[2]];
      if (!(nd.flags & LIBXSMM_MELTW_FLAG_TERNARY_REUSE_IN_2_AS_OUT) || Cn.m != m || Cn.n != n) return false;
      nd.ld = Cn.ld; nd.type = Cn.type;
    }
    return true;
  }
  const EqnNode& l = e.nodes[nd.child[0]];
  nd.type = nd.dtype;
  if (nd.kind == EQ_UNARY) {
    if (is_reduce(nd.op)) {
      if (nd.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS) { nd.m = l.n; nd.n = 1; nd.ld = l.n; }
      else if (nd.flags & LIBXSMM_MELTW_FLAG_UNARY_REDUCE_COLS) { nd.m = l.m; nd.n = 1; nd.ld = l.m; }
      else return false;
    } else { nd.m = l.m; nd.n = l.n; nd.ld = l.m; }
  } else if (nd.kind == EQ_BINARY) {
    const EqnNode& r = e.nodes[nd.child[1]];
    nd.m = std::max(l.m, r.m); nd.n = std::max(l.n, r.n); nd.ld = nd.m;
    if (nd.op == LIBXSMM_MELTW_TYPE_BINARY_ZIP) nd.type = LIBXSMM_DATATYPE_F32;       // two 16-bit halves -> one f32 [ref: mateltwise ref ZIP]
    if (nd.op == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD) {          // a dot product: one value [ref: mateltwise ref :2523-2542]
      if (l.m != r.m || l.n != r.n) return false;
      nd.m = nd.n = nd.ld = 1;
    }
  } else {
    const EqnNode& r = e.nodes[nd.child[1]]; const EqnNode& r2 = e.nodes[nd.child[2]];
    nd.m = std::max(r2.m, std::max(l.m, r.m)); nd.n = std::max(r2.n, std::max(l.n, r.n)); nd.ld = nd.m;
  }
  return true;
}

void postorder(const Equation& e, int id, std::vector<int>& out) {
  const EqnNode& nd = e.nodes[id];
  if (nd.kind == EQ_ARG) return;
  for (int c = 0; c < arity(nd.kind); ++c) postorder(e, nd.child[c], out);
  out.push_back(id);
}

void print_node(const Equation& e, int id, int depth) {
  const EqnNode& nd = e.nodes[id];
  for (int i = 0; i < depth; ++i) std::printf("  ");
  if (nd.kind == EQ_ARG) std::printf("ARG %d (%dx%d ld %d type %d)\n", nd.in_pos, nd.m, nd.n, nd.ld, nd.type);
  else {
    std::printf("%s op %d flags %u dtype %d\n", nd.kind == EQ_UNARY ? "UNARY" : nd.kind == EQ_BINARY ? "BINARY" : "TERNARY", nd.op, nd.flags, nd.dtype);
    for (int c = 0; c < arity(nd.kind); ++c) if (nd.child[c] >= 0) print_node(e, nd.child[c], depth + 1);
  }
}
void print_rpn(const Equation& e, int id) {
  const EqnNode& nd = e.nodes[id];
  for (int c = 0; c < arity(nd.kind); ++c) if (nd.child[c] >= 0) print_rpn(e, nd.child[c]);
  if (nd.kind == EQ_ARG) std::printf("ARG%d ", nd.in_pos); else std::printf("%s%d ", nd.kind == EQ_UNARY ? "U" : nd.kind == EQ_BINARY ? "B" : "T", nd.op);
}


// ---- whole-tree fusion for element-wise equations ---------------------------------------------------------------------
// Conditions: every op is element-wise arithmetic with an f32 op type, every broadcast operand is an argument, all full
// operands share the output's m x n, m and every leading dimension are multiples of 8.  One thread = 8 consecutive rows
// of one column; the expression is emitted in post-order on 8-element register arrays with the SAME per-element
// formulas as meltw_kernels.hip (contraction off), so the result is bit-identical to the step chain.
const char* kFusedPrelude = R"SRC(
#define GM __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float daz(float x) { return ((__float_as_uint(x) & 0x7f800000u) == 0u) ? __uint_as_float(__float_as_uint(x) & 0x80000000u) : x; }
__device__ __forceinline__ unsigned int f2bf_pk(float lo, float hi) { const f32x2 v = {daz(lo), daz(hi)}; return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ void ld_f32(float (&x)[8], GM const float* p) { const f32x4 a = *(GM const f32x4*)p, b = *(GM const f32x4*)(p + 4);
  x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3]; }
__device__ __forceinline__ void ld_bf16(float (&x)[8], GM const unsigned short* p) { const u32x4 v = *(GM const u32x4*)p;
  _Pragma("unroll") for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(v[e] << 16); x[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); } }
__device__ __forceinline__ void st_f32(GM float* p, const float (&y)[8]) { f32x4 a, b; a[0] = y[0]; a[1] = y[1]; a[2] = y[2]; a[3] = y[3]; b[0] = y[4]; b[1] = y[5]; b[2] = y[6]; b[3] = y[7];
  *(GM f32x4*)p = a; *(GM f32x4*)(p + 4) = b; }
__device__ __forceinline__ void st_bf16(GM unsigned short* p, const float (&y)[8]) { u32x4 v; _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = f2bf_pk(y[2 * e], y[2 * e + 1]); *(GM u32x4*)p = v; }
__device__ __forceinline__ float sigm(float x) { return (tanhf(x * 0.5f) + 1.0f) * 0.5f; }
)SRC";

const char* unary_expr(int t) {     // %s = operand, A = alpha  [same formulas as unary_math in meltw_kernels.hip]
  switch (t) {
    case LIBXSMM_MELTW_TYPE_UNARY_IDENTITY: return "X";
    case LIBXSMM_MELTW_TYPE_UNARY_X2: return "X * X";
    case LIBXSMM_MELTW_TYPE_UNARY_SQRT: return "sqrtf(X)";
    case LIBXSMM_MELTW_TYPE_UNARY_RELU: return "(X <= 0.0f) ? 0.0f : X";
    case LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU: return "(X <= 0.0f) ? A * X : X";
    case LIBXSMM_MELTW_TYPE_UNARY_TANH: return "tanhf(X)";
    case LIBXSMM_MELTW_TYPE_UNARY_SIGMOID: return "sigm(X)";
    case LIBXSMM_MELTW_TYPE_UNARY_NEGATE: return "-1.0f * X";
    case LIBXSMM_MELTW_TYPE_UNARY_INC: return "X + 1.0f";
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL: return "1.0f / X";
    case LIBXSMM_MELTW_TYPE_UNARY_RECIPROCAL_SQRT: return "1.0f / sqrtf(X)";
    case LIBXSMM_MELTW_TYPE_UNARY_EXP: return "expf(X)";
    default: return nullptr;
  }
}
const char* binary_expr(int t) {
  switch (t) {
    case LIBXSMM_MELTW_TYPE_BINARY_ADD: return "X + Y";
    case LIBXSMM_MELTW_TYPE_BINARY_SUB: return "X - Y";
    case LIBXSMM_MELTW_TYPE_BINARY_MUL: return "X * Y";
    case LIBXSMM_MELTW_TYPE_BINARY_DIV: return "X / Y";
    case LIBXSMM_MELTW_TYPE_BINARY_MAX: return "(X > Y) ? X : Y";
    case LIBXSMM_MELTW_TYPE_BINARY_MIN: return "(X > Y) ? Y : X";
    default: return nullptr;
  }
}
std::string subst(const char* tmpl, const std::string& x, const std::string& y, const std::string& alpha) {
  std::string out;
  for (const char* c = tmpl; *c; ++c) { if (*c == 'X') out += x; else if (*c == 'Y') out += y; else if (*c == 'A' && (c == tmpl || c[-1] == ' ') ) out += alpha; else out += *c; }
  return out;
}

int bcast_of(const EqnNode& parent, int operand) {   // 0 none, 1 row, 2 col, 3 scalar
  const unsigned int f = parent.flags;
  if (parent.kind == EQ_UNARY) { if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW) return 1; if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL) return 2; if (f & LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR) return 3; return 0; }
  if (parent.kind == EQ_BINARY) {
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_ROW_IN_0 << operand)) return 1; if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_COL_IN_0 << operand)) return 2;
    if (f & (LIBXSMM_MELTW_FLAG_BINARY_BCAST_SCALAR_IN_0 << operand)) return 3; return 0;
  }
  if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_ROW_IN_0 << operand)) return 1; if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_COL_IN_0 << operand)) return 2;
  if (f & (LIBXSMM_MELTW_FLAG_TERNARY_BCAST_SCALAR_IN_0 << operand)) return 3; return 0;
}

// A reduction (one node, or a REDUCE_COLS / REDUCE_ROWS pair) that folds an element-wise M x N operand into ONE number: the max and the sum
// of a softmax, the sum of a softmax backward pass [ref: samples/equation/equation_softmax.c:527-538,676-688]
struct ScalarReduce { int src; int fold; bool square; };     // fold: 0 add, 1 max, 2 min
bool scalar_reduce(const Equation& e, int id, int M, int N, ScalarReduce& r) {
  const EqnNode& nd = e.nodes[id];
  const auto kind = [](int op, int& fold, bool& sq) {
    switch (op) {
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD: fold = 0; sq = false; return true;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD: fold = 0; sq = true; return true;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX: fold = 1; sq = false; return true;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN: fold = 2; sq = false; return true;
      default: return false;
    }
  };
  const unsigned int dirs = LIBXSMM_MELTW_FLAG_UNARY_REDUCE_ROWS | LIBXSMM_MELTW_FLAG_UNARY_REDUCE_COLS;
  if (nd.kind != EQ_UNARY || nd.m != 1 || nd.n != 1 || nd.dtype != LIBXSMM_DATATYPE_F32 || !kind(nd.op, r.fold, r.square) || (nd.flags & ~dirs) || !(nd.flags & dirs)) return false;
  r.src = nd.child[0];
  const EqnNode& c = e.nodes[nd.child[0]];
  if (c.kind == EQ_UNARY && is_reduce(c.op)) {
    int f2 = 0; bool s2 = false;
    if (c.dtype != LIBXSMM_DATATYPE_F32 || !kind(c.op, f2, s2) || f2 != r.fold || r.square || (c.flags & ~dirs) || !(c.flags & dirs)) return false;
    r.square = s2; r.src = c.child[0];
  }
  return e.nodes[r.src].m == M && e.nodes[r.src].n == N;
}

// returns false when the tree is not fusable; on success `src` holds the kernel source.
// Two forms.  Element-wise trees: a grid of threads, one 8-row unit each.  Trees with reductions to ONE number inside (or as the head): one
// workgroup of 256 threads that walks the units once per reduction ("phase"), folds its partial results in LDS and carries the number in a
// register into the phases that broadcast it -- the operand trees are re-evaluated from the (cache-resident) arguments, never written.  Unit t
// belongs to thread t % 256 in every phase, so what a DUMP node wrote is read back (as an argument) by the thread that wrote it.
bool generate_fused(const Equation& e, const libxsmm_meqn_arg_shape& out, int eqn_idx, std::string& src, std::string& fname, EqnPlan& plan, long long& total) {
  const EqnNode& root = e.nodes[0];
  const bool scalar_root = root.m == 1 && root.n == 1;
  int M = root.m, N = root.n;                      // the extent of the element-wise part: the head's, or (a head that is one number) the operands'
  if (scalar_root) for (const EqnNode& nd : e.nodes) { M = std::max(M, nd.m); N = std::max(N, nd.n); }
  if (M % 8 != 0 || (out.type != LIBXSMM_DATATYPE_F32 && out.type != LIBXSMM_DATATYPE_BF16)) return false;
  if (!scalar_root && (root.m != M || root.n != N || out.ld % 8 != 0)) return false;
  const long long units = (long long)(M / 8) * N;
  char buf[768];
  std::string phases;
  std::map<int, int> arg_slot;                     // input position -> kernel argument index
  std::vector<std::pair<int, int> > arg_types;     // (input position, datatype)
  const auto slot_of_arg = [&](const EqnNode& ch) -> int {
    if (ch.type != LIBXSMM_DATATYPE_F32 && ch.type != LIBXSMM_DATATYPE_BF16) return -1;
    if (arg_slot.find(ch.in_pos) == arg_slot.end()) { arg_slot[ch.in_pos] = (int)arg_types.size(); arg_types.push_back({ch.in_pos, ch.type}); }
    else if (arg_types[arg_slot[ch.in_pos]].second != ch.type) return -1;
    return arg_slot[ch.in_pos];
  };
  std::function<bool(int, int, std::string&, std::string&)> value;     // (node, broadcast kind, name, code of the enclosing unit loop)
  std::function<bool(int, std::string&)> elem, scalar_value;           // element-wise op node -> v<id>[8]; one-number subtree -> s<id>
  std::function<bool(int, std::string&)> vector_value;                 // REDUCE_COLS subtree (one number per row) -> LDS array w<id>[M]
  // Vector-valued reductions inside a tree (a sum over the columns, broadcast back along them) as a phase of the one-workgroup kernel.  Measured in
  // round 3 (tools/bench_meqn_vecred.py, profiles/r03_meqn_vecred.jsonl): x * colsum(x^2) and a column softmax as ONE kernel against the chain of TPP
  // launches -- 64 x 128: 7.9 vs 19.4 us and 58 vs 109 us (the launches of the chain dominate); 64 x 1024: 56 vs 14.6 us and 457 vs 81 us, 256 x 512:
  // 40 vs 10.8 us (one workgroup walks what the chain spreads over the chip).  So: on up to 2^14 elements by default;
  // LIBXSMM_HIP_MEQN_VECRED=1 always, =0 never.
  static const int vecred_mode = []() { const char* v = getenv("LIBXSMM_HIP_MEQN_VECRED"); return (v && (v[0] == '0' || v[0] == '1')) ? v[0] - '0' : 2; }();
  const bool vecred = vecred_mode == 1 || (vecred_mode == 2 && (long long)M * (long long)N <= (1ll << 14));
  std::string lds_decl;
  int n_vec = 0;
  const auto operand = [&](const EqnNode& parent, int c, std::string& name, std::string& body) -> bool { return value(parent.child[c], bcast_of(parent, c), name, body); };
  const std::string unit_loop = "  for (long long t = threadIdx.x; t < " + std::to_string(units) + "LL; t += 256) {\n  const long long j = t / " + std::to_string(M / 8) + ", i = (t - j * " + std::to_string(M / 8) + ") * 8;\n";

  value = [&](int id, int bc, std::string& name, std::string& body) -> bool {
    const EqnNode& ch = e.nodes[id];
    if (ch.kind != EQ_ARG) {
      if (ch.m == 1 && ch.n == 1) {     // a number computed by an earlier phase, broadcast
        std::string s;
        if (bc != 3 || !scalar_value(id, s)) return false;
        name = "b" + std::to_string(id);
        body += "  float " + name + "[8]; _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + name + "[e] = " + s + ";\n";
        return true;
      }
      if (bc == 2 && vecred && ch.m == M && ch.n == 1 && N > 1) {     // one number per row from an earlier phase, broadcast along the columns
        std::string w;
        if (!vector_value(id, w)) return false;
        name = "b" + std::to_string(id);
        body += "  float " + name + "[8]; _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + name + "[e] = " + w + "[i + e];\n";
        return true;
      }
      if (bc != 0 || ch.m != M || ch.n != N || !elem(id, body)) return false;
      name = "v" + std::to_string(id);
      return true;
    }
    if (bc == 0 && (ch.m != M || ch.n != N || ch.ld % 8 != 0)) return false;
    if (bc == 2 && ch.m != M) return false;
    const int k = slot_of_arg(ch);
    if (k < 0) return false;
    name = "a" + std::to_string(id);
    const char* T = ch.type == LIBXSMM_DATATYPE_F32 ? "float" : "unsigned short";
    const char* LD = ch.type == LIBXSMM_DATATYPE_F32 ? "ld_f32" : "ld_bf16";
    if (bc == 0 || bc == 2) {
      std::snprintf(buf, sizeof(buf), "  float %s[8]; %s(%s, (GM const %s*)in%d + i%s);\n", name.c_str(), LD, name.c_str(), T, k,
                    bc == 0 ? (" + j * " + std::to_string(ch.ld) + "LL").c_str() : "");
    } else {
      const std::string idx = bc == 1 ? ("j * " + std::to_string(ch.ld) + "LL") : std::string("0");
      if (ch.type == LIBXSMM_DATATYPE_F32) std::snprintf(buf, sizeof(buf), "  float %s[8]; { const float s = ((GM const float*)in%d)[%s]; _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) %s[e] = s; }\n", name.c_str(), k, idx.c_str(), name.c_str());
      else std::snprintf(buf, sizeof(buf), "  float %s[8]; { const float s = __uint_as_float((unsigned int)((GM const unsigned short*)in%d)[%s] << 16); _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) %s[e] = s; }\n", name.c_str(), k, idx.c_str(), name.c_str());
    }
    body += buf;
    return true;
  };

  elem = [&](int id, std::string& body) -> bool {
    const EqnNode& nd = e.nodes[id];
    if (nd.dtype != LIBXSMM_DATATYPE_F32 || nd.m != M || nd.n != N) return false;
    std::string x, y, z, alpha = "0.0f";
    const std::string v = "v" + std::to_string(id);
    if (nd.kind == EQ_UNARY) {
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_DUMP) {      // identity that also lands in ops_args[pos].primary: an f32 M x N image, ld = M, below the head [ref: matequation ref :58-60]
        if (id == 0 || nd.flags != 0 || nd.op_arg_pos < 0 || plan.fused_dumps.size() >= 4 || !operand(nd, 0, x, body)) return false;
        const std::string d = "dump" + std::to_string(plan.fused_dumps.size());
        plan.fused_dumps.push_back(nd.op_arg_pos);
        body += "  float " + v + "[8];\n  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + v + "[e] = " + x + "[e];\n  st_f32((GM float*)" + d + " + i + j * " + std::to_string(M) + "LL, " + v + ");\n";
        return true;
      }
      const char* t = unary_expr(nd.op);
      if (!t || (nd.flags & ~(unsigned int)(LIBXSMM_MELTW_FLAG_UNARY_BCAST_ROW | LIBXSMM_MELTW_FLAG_UNARY_BCAST_COL | LIBXSMM_MELTW_FLAG_UNARY_BCAST_SCALAR))) return false;
      if (!operand(nd, 0, x, body)) return false;
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU) { if (nd.op_arg_pos < 0 || plan.fused_alphas.size() >= 8) return false; alpha = "alpha" + std::to_string(plan.fused_alphas.size()); plan.fused_alphas.push_back(nd.op_arg_pos); }
      body += "  float " + v + "[8];\n  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + v + "[e] = " + subst(t, x + "[e]", "", alpha) + ";\n";
    } else if (nd.kind == EQ_BINARY) {
      const char* t = binary_expr(nd.op);
      if (!t || !operand(nd, 0, x, body) || !operand(nd, 1, y, body)) return false;
      body += "  float " + v + "[8];\n  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + v + "[e] = " + subst(t, x + "[e]", y + "[e]", alpha) + ";\n";
    } else {
      if (nd.op != LIBXSMM_MELTW_TYPE_TERNARY_MULADD && nd.op != LIBXSMM_MELTW_TYPE_TERNARY_NMULADD) return false;
      if (!operand(nd, 0, x, body) || !operand(nd, 1, y, body) || !operand(nd, 2, z, body)) return false;
      body += "  float " + v + "[8];\n  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) { const float prod = " + x + "[e] * " + (nd.op == LIBXSMM_MELTW_TYPE_TERNARY_MULADD ? y : z) + "[e]; " + v + "[e] = " +
              (nd.op == LIBXSMM_MELTW_TYPE_TERNARY_MULADD ? z + "[e] + prod" : y + "[e] - prod") + "; }\n";
    }
    return true;
  };

  scalar_value = [&](int id, std::string& name) -> bool {
    const EqnNode& nd = e.nodes[id];
    name = "s" + std::to_string(id);
    if (nd.m != 1 || nd.n != 1) return false;
    if (nd.kind == EQ_ARG) {
      const int k = slot_of_arg(nd);
      if (k < 0) return false;
      phases += nd.type == LIBXSMM_DATATYPE_F32 ? "  const float " + name + " = ((GM const float*)in" + std::to_string(k) + ")[0];\n"
                                                : "  const float " + name + " = __uint_as_float((unsigned int)((GM const unsigned short*)in" + std::to_string(k) + ")[0] << 16);\n";
      return true;
    }
    if (nd.dtype != LIBXSMM_DATATYPE_F32) return false;
    ScalarReduce r; r.src = -1; r.fold = 0; r.square = false;
    const bool dot = nd.kind == EQ_BINARY && nd.op == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD && nd.flags == 0;
    if (dot || scalar_reduce(e, id, M, N, r)) {
      std::string body, x, y, fold;
      if (dot) {
        if (e.nodes[nd.child[0]].m != M || e.nodes[nd.child[0]].n != N || !value(nd.child[0], 0, x, body) || !value(nd.child[1], 0, y, body)) return false;
        fold = "{ const float prod = " + x + "[e] * " + y + "[e]; acc = acc + prod; }";
      } else {
        if (!value(r.src, 0, x, body)) return false;
        fold = r.fold == 1 ? "acc = (acc < " + x + "[e]) ? " + x + "[e] : acc;" : r.fold == 2 ? "acc = (acc > " + x + "[e]) ? " + x + "[e] : acc;"
             : r.square ? "{ const float sq = " + x + "[e] * " + x + "[e]; acc = acc + sq; }" : "acc = acc + " + x + "[e];";
      }
      const char* init = r.fold == 1 ? "-3.402823466e+38f" : r.fold == 2 ? "3.402823466e+38f" : "0.0f";      // [ref: mateltwise ref :1386,:1415]
      const char* combine = r.fold == 1 ? "(acc < o) ? o : acc" : r.fold == 2 ? "(acc > o) ? o : acc" : "acc + o";
      phases += "  float " + name + ";\n  { float acc = " + init + ";\n" + unit_loop + body + "  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + fold + "\n  }\n"
                "  part[threadIdx.x] = acc; __syncthreads();\n"
                "  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { const float o = part[threadIdx.x + s]; acc = " + combine + "; part[threadIdx.x] = acc; } __syncthreads(); }\n"
                "  " + name + " = part[0]; __syncthreads(); }\n";
      return true;
    }
    // arithmetic on numbers (the reciprocal of a sum, ...): every operand is a number itself, broadcast flags say nothing new
    std::string x, y, z;
    if (nd.kind == EQ_UNARY) {
      const char* t = unary_expr(nd.op);
      if (!t || nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || !scalar_value(nd.child[0], x)) return false;
      phases += "  const float " + name + " = " + subst(t, x, "", "0.0f") + ";\n";
    } else if (nd.kind == EQ_BINARY) {
      const char* t = binary_expr(nd.op);
      if (!t || !scalar_value(nd.child[0], x) || !scalar_value(nd.child[1], y)) return false;
      phases += "  const float " + name + " = " + subst(t, x, y, "0.0f") + ";\n";
    } else {
      if ((nd.op != LIBXSMM_MELTW_TYPE_TERNARY_MULADD && nd.op != LIBXSMM_MELTW_TYPE_TERNARY_NMULADD) || !scalar_value(nd.child[0], x) || !scalar_value(nd.child[1], y) || !scalar_value(nd.child[2], z)) return false;
      phases += nd.op == LIBXSMM_MELTW_TYPE_TERNARY_MULADD ? "  const float " + name + " = " + z + " + " + x + " * " + y + ";\n" : "  const float " + name + " = " + y + " - " + x + " * " + z + ";\n";
    }
    return true;
  };

  // REDUCE_COLS of an element-wise M x N operand (ADD, X2-ADD, MAX, MIN): thread r owns the 8-row blocks r, r + 256, ... and walks the columns in
  // ascending order -- the reference's order [ref: mateltwise ref :1065-1130], so a plain sum is the same chain of additions -- then parks its
  // 8 results in LDS; one barrier makes them visible to the phases that broadcast them.
  vector_value = [&](int id, std::string& name) -> bool {
    const EqnNode& nd = e.nodes[id];
    name = "w" + std::to_string(id);
    if (nd.kind != EQ_UNARY || nd.dtype != LIBXSMM_DATATYPE_F32 || nd.m != M || nd.n != 1 || nd.flags != LIBXSMM_MELTW_FLAG_UNARY_REDUCE_COLS) return false;
    int fold = 0; bool sq = false;
    switch (nd.op) {
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD: break;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD: sq = true; break;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX: fold = 1; break;
      case LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN: fold = 2; break;
      default: return false;
    }
    const EqnNode& srcn = e.nodes[nd.child[0]];
    if (srcn.m != M || srcn.n != N || M > 2048 || n_vec >= 4) return false;
    std::string body, x;
    const size_t dumps_before = plan.fused_dumps.size();
    if (!value(nd.child[0], 0, x, body)) return false;
    if (plan.fused_dumps.size() != dumps_before) return false;      // a DUMP image written here would be read back by another thread later
    ++n_vec;
    lds_decl += "  __shared__ float " + name + "[" + std::to_string(M) + "];\n";
    const char* init = fold == 1 ? "-3.402823466e+38f" : fold == 2 ? "3.402823466e+38f" : "0.0f";
    const std::string step = fold == 1 ? "acc[e] = (acc[e] < " + x + "[e]) ? " + x + "[e] : acc[e];" : fold == 2 ? "acc[e] = (acc[e] > " + x + "[e]) ? " + x + "[e] : acc[e];"
                           : sq ? "{ const float sq = " + x + "[e] * " + x + "[e]; acc[e] = acc[e] + sq; }" : "acc[e] = acc[e] + " + x + "[e];";
    phases += "  for (long long ib = threadIdx.x; ib < " + std::to_string(M / 8) + "LL; ib += 256) {\n  const long long i = ib * 8;\n  float acc[8]; _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) acc[e] = " + init + ";\n"
              "  for (long long j = 0; j < " + std::to_string(N) + "LL; ++j) {\n" + body + "  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + step + "\n  }\n"
              "  _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) " + name + "[i + e] = acc[e];\n  }\n  __syncthreads();\n";
    return true;
  };

  std::string body, head;
  if (scalar_root) { if (M == 1 && N == 1) return false; if (!scalar_value(0, head)) return false; }
  else if (!elem(0, body)) return false;
  const bool phased = scalar_root || !phases.empty();
  if (phased && units > 256 * 64) return false;      // one workgroup: beyond ~10^5 elements the chain of full-grid kernels is the faster form
  if (arg_types.empty() || arg_types.size() > 24) return false;
  total = phased ? 256 : units;
  fname = std::string(phased ? "meqn_jit_r" : "meqn_jit_e") + std::to_string(eqn_idx) + "_" + std::to_string(M) + "x" + std::to_string(N) + "_o" + std::to_string((int)out.type);
  src = kFusedPrelude;
  src += "extern \"C\" __global__ __launch_bounds__(256) void " + fname + "(";
  for (size_t k = 0; k < arg_types.size(); ++k) src += "const void* in" + std::to_string(k) + ", ";
  src += "void* out";
  for (size_t k = 0; k < plan.fused_alphas.size(); ++k) src += ", float alpha" + std::to_string(k);
  for (size_t k = 0; k < plan.fused_dumps.size(); ++k) src += ", void* dump" + std::to_string(k);
  src += ") {\n";
  const std::string store = std::string("  ") + (out.type == LIBXSMM_DATATYPE_F32 ? "st_f32((GM float*)" : "st_bf16((GM unsigned short*)") + "out + i + j * " + std::to_string((int)out.ld) + "LL, v0);\n";
  if (phased) {
    src += "  __shared__ float part[256];\n" + lds_decl + phases;
    if (scalar_root) src += out.type == LIBXSMM_DATATYPE_F32 ? "  if (threadIdx.x == 0) *(GM float*)out = " + head + ";\n}\n"
                                                             : "  if (threadIdx.x == 0) *(GM unsigned short*)out = (unsigned short)(f2bf_pk(" + head + ", 0.0f) & 0xffffu);\n}\n";
    else src += unit_loop + body + store + "  }\n}\n";
  } else {
    std::snprintf(buf, sizeof(buf), "  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;\n  if (t >= %lldLL) return;\n  const long long j = t / %d, i = (t - j * %d) * 8;\n", units, M / 8, M / 8);
    src += buf;
    src += body + store + "}\n";
  }
  for (auto& a : arg_types) {
    plan.fused_inputs.push_back(a.first);
    char scalar = 0;
    for (const EqnNode& nd : e.nodes) if (nd.kind == EQ_ARG && nd.in_pos == a.first && nd.m == 1 && nd.n == 1) scalar = 1;
    plan.fused_scalar.push_back(scalar);
  }
  return true;
}

}  // namespace

const char* meqn_plan_name(const EqnPlan* plan) { return (plan && plan->fused) ? jit_name(plan->fused) : "meqn_tpp_chain"; }
void free_meqn_plan(EqnPlan* plan) { if (plan && plan->fused) jit_release(plan->fused); delete plan; }
// libxsmm_finalize: the equation objects (and their handle caches, whose handles the runtime has just released) go with the registry
void free_meqn_equations() {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  for (Equation* e : g_eqns) delete e;
  g_eqns.clear();
}

void run_meqn(EqnPlan* plan, const void* param) {
  const libxsmm_meqn_param* p = (const libxsmm_meqn_param*)param;
  if (!p->inputs || !p->output.primary) { set_error(-2, "matrix equation called without inputs / output"); return; }
  if (plan->fused && jit_on_current_device(plan->fused)) {
    const void* ptrs[24]; float alphas[8]; void* args[37]; int na = 0; bool ok = true;
    rt_scratch_reset();
    for (size_t i = 0; i < plan->fused_inputs.size(); ++i) {
      ptrs[i] = p->inputs[plan->fused_inputs[i]].primary;
      if (ptrs[i] && plan->fused_scalar[i]) ptrs[i] = rt_small_host_input(ptrs[i], 8);
      ok = ok && ptrs[i] && (plan->fused_scalar[i] || (((size_t)ptrs[i]) & 15) == 0);
      args[na++] = (void*)&ptrs[i];
    }
    void* outp = p->output.primary;
    if (plan->out_scalar) { outp = rt_small_host_output(outp, plan->out_scalar_bytes); if (!outp) return; }
    else ok = ok && (((size_t)outp) & 15) == 0;
    args[na++] = (void*)&outp;
    for (size_t i = 0; i < plan->fused_alphas.size() && ok; ++i) {
      if (!p->ops_args || !p->ops_args[plan->fused_alphas[i]].primary) { set_error(-2, "matrix equation: op argument %d is NULL", plan->fused_alphas[i]); return; }
      alphas[i] = *(const float*)p->ops_args[plan->fused_alphas[i]].primary; args[na++] = (void*)&alphas[i];
    }
    void* dumps[4];
    for (size_t i = 0; i < plan->fused_dumps.size() && ok; ++i) {
      if (!p->ops_args || !p->ops_args[plan->fused_dumps[i]].primary) { set_error(-2, "matrix equation: DUMP destination (op argument %d) is NULL", plan->fused_dumps[i]); return; }
      dumps[i] = p->ops_args[plan->fused_dumps[i]].primary; ok = ok && (((size_t)dumps[i]) & 15) == 0; args[na++] = (void*)&dumps[i];
    }
    if (ok) { rt_finish_launch(jit_launch(plan->fused, args, rt_stream()), "meqn_jit"); return; }
  }
  rt_scratch_reset();
  char* const out_primary = plan->out_scalar ? (char*)rt_small_host_output(p->output.primary, plan->out_scalar_bytes) : (char*)p->output.primary;
  if (!out_primary) return;
  char* ws = nullptr;
  if (plan->nslots > 0) { ws = (char*)rt_workspace(plan->slot_bytes * (size_t)plan->nslots + (plan->has_gemm ? ((size_t)8 << 20) : 0)); if (!ws) return; }
  const char* kname = nullptr;
  int err = 0;
  for (size_t s = 0; s < plan->steps.size() && err == 0; ++s) {
    const EqnStep& st = plan->steps[s];
    MeltwArgs a = st.args;
    const char* src[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < 3; ++c) {
      if (st.src[c] == INT32_MIN) continue;
      src[c] = st.src[c] >= 0 ? (const char*)p->inputs[st.src[c]].primary : ws + plan->slot_bytes * (size_t)(-st.src[c] - 1);
      if (!src[c]) { set_error(-2, "matrix equation: input %d is NULL", st.src[c]); return; }
      // a 1 x 1 argument is a scalar the caller typically keeps on its stack (learning rate, mean, ...): staged when it is host memory
      if (st.src[c] >= 0 && st.scalar_arg[c]) { src[c] = (const char*)rt_small_host_input(src[c], 8); if (!src[c]) return; }
    }
    a.in0 = src[0]; a.in1 = src[1]; a.in2 = src[2];
    a.out = st.out_loc == INT32_MIN ? out_primary : (st.out_loc >= 0 ? (char*)p->inputs[st.out_loc].primary : ws + plan->slot_bytes * (size_t)(-st.out_loc - 1));
    if (st.skip_if_in_place && (const char*)a.out == src[0]) continue;     // C += A * B with the accumulator passed as the output: nothing to copy
    if (st.gemm) {   // a MATMUL / BRGEMM node: the dense kernel, called like any other handle (its own launch bookkeeping included)
      libxsmm_gemm_param gp; std::memset(&gp, 0, sizeof(gp));
      unsigned long long blocks = 1;
      gp.a.primary = (void*)src[0]; gp.b.primary = (void*)src[1]; gp.c.primary = a.out;
      if (st.br_from_op >= 0) {
        if (!p->ops_args || !p->ops_args[st.br_from_op].tertiary) { set_error(-2, "matrix equation: BRGEMM block count (op argument %d, tertiary) is NULL", st.br_from_op); return; }
        blocks = *(const unsigned long long*)p->ops_args[st.br_from_op].tertiary; gp.op.tertiary = &blocks;
      }
      rt_workspace_reserve(plan->slot_bytes * (size_t)plan->nslots);     // the kernel's own partial-sum workspace goes behind the slots
      rt_nest(+1);          // keeps this call's staged scalars / pending copy-backs alive and defers the sync to the end of the chain
      ((libxsmm_gemmfunction)st.gemm)(&gp);
      rt_nest(-1);
      rt_workspace_reserve(0);
      kname = nullptr;
      continue;
    }
    if (st.alpha_from_op >= 0) {
      if (!p->ops_args || !p->ops_args[st.alpha_from_op].primary) { set_error(-2, "matrix equation: op argument %d is NULL", st.alpha_from_op); return; }
      a.scalar_f32 = *(const float*)p->ops_args[st.alpha_from_op].primary;
    }
    if (st.root_side == 1) {
      if (!p->output.secondary) { set_error(-2, "matrix equation: the head is a ReLU with bitmask but output.secondary is NULL"); return; }
      a.aux_out = p->output.secondary;
    } else if (st.root_side == 2) {
      if (!p->output.secondary) { set_error(-2, "matrix equation: the head is UNZIP but output.secondary (byte offset of the upper halves) is NULL"); return; }
      a.scalar_u64 = *(const unsigned long long*)p->output.secondary;
    } else if (st.root_side == 3) {
      if (!p->output.secondary) { set_error(-2, "matrix equation: the head is SCATTER but output.secondary (the index list) is NULL"); return; }
      a.aux_out = p->output.secondary;
    }
    if (st.idx_from_input >= 0) {
      if (!p->inputs[st.idx_from_input].secondary) { set_error(-2, "matrix equation: GATHER needs its index list in inputs[%d].secondary", st.idx_from_input); return; }
      a.aux_in = p->inputs[st.idx_from_input].secondary;
    }
    if (st.dump_from_op >= 0) {
      if (!p->ops_args || !p->ops_args[st.dump_from_op].primary) { set_error(-2, "matrix equation: DUMP destination (op argument %d) is NULL", st.dump_from_op); return; }
      a.aux_out = p->ops_args[st.dump_from_op].primary;
    }
    err = launch_meltw(a, rt_stream(), &kname);
  }
  rt_finish_launch(err, kname ? kname : "meqn");
}

}  // namespace xamd

using namespace xamd;

extern "C" {

LIBXSMM_API libxsmm_blasint libxsmm_meqn_create(void) {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  g_eqns.push_back(new Equation());
  return (libxsmm_blasint)g_eqns.size() - 1;
}
LIBXSMM_API libxsmm_meqn_arg_shape libxsmm_create_meqn_arg_shape(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint ld, libxsmm_datatype type) {
  libxsmm_meqn_arg_shape s; s.m = m; s.n = n; s.ld = ld; s.type = type; return s;
}
LIBXSMM_API libxsmm_matrix_arg_attributes libxsmm_create_matrix_arg_attributes(libxsmm_matrix_arg_type type, libxsmm_matrix_arg_set_type set_type,
  libxsmm_blasint set_cardinality_hint, libxsmm_blasint set_stride_hint) {
  libxsmm_matrix_arg_attributes a; a.type = type; a.set_type = set_type; a.set_cardinality_hint = set_cardinality_hint; a.set_stride_hint = set_stride_hint; return a;
}
LIBXSMM_API libxsmm_meqn_arg_metadata libxsmm_create_meqn_arg_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint in_arg_pos) {
  libxsmm_meqn_arg_metadata m; m.eqn_idx = eqn_idx; m.in_arg_pos = in_arg_pos; return m;
}
LIBXSMM_API libxsmm_meqn_op_metadata libxsmm_create_meqn_op_metadata(libxsmm_blasint eqn_idx, libxsmm_blasint op_arg_pos) {
  libxsmm_meqn_op_metadata m; m.eqn_idx = eqn_idx; m.op_arg_pos = op_arg_pos; return m;
}
LIBXSMM_API int libxsmm_meqn_push_back_arg(libxsmm_meqn_arg_metadata md, libxsmm_meqn_arg_shape shape, libxsmm_matrix_arg_attributes attr) {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  Equation* e = get(md.eqn_idx);
  if (!e) return 1;
  EqnNode nd; nd.kind = EQ_ARG; nd.in_pos = md.in_arg_pos; nd.m = shape.m; nd.n = shape.n; nd.ld = shape.ld; nd.type = shape.type;
  nd.set = (attr.type == LIBXSMM_MATRIX_ARG_TYPE_SET) ? 1 : 0; nd.set_type = (int)attr.set_type; nd.set_stride = attr.set_stride_hint;
  return push(*e, nd);
}
static int push_op(libxsmm_meqn_op_metadata md, int kind, int type, libxsmm_datatype dtype, libxsmm_bitfield flags) {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  Equation* e = get(md.eqn_idx);
  if (!e) return 1;
  EqnNode nd; nd.kind = kind; nd.op = type; nd.dtype = dtype; nd.flags = (unsigned int)flags; nd.op_arg_pos = md.op_arg_pos;
  return push(*e, nd);
}
LIBXSMM_API int libxsmm_meqn_push_back_unary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_unary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) {
  return push_op(md, EQ_UNARY, (int)type, dtype, flags);
}
LIBXSMM_API int libxsmm_meqn_push_back_binary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_binary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) {
  return push_op(md, EQ_BINARY, (int)type, dtype, flags);
}
LIBXSMM_API int libxsmm_meqn_push_back_ternary_op(libxsmm_meqn_op_metadata md, libxsmm_meltw_ternary_type type, libxsmm_datatype dtype, libxsmm_bitfield flags) {
  return push_op(md, EQ_TERNARY, (int)type, dtype, flags);
}
LIBXSMM_API void libxsmm_meqn_tree_print(libxsmm_blasint idx) {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  Equation* e = get(idx);
  if (e && !e->nodes.empty()) print_node(*e, 0, 0);
}
LIBXSMM_API void libxsmm_meqn_rpn_print(libxsmm_blasint idx) {
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  Equation* e = get(idx);
  if (e && !e->nodes.empty()) { print_rpn(*e, 0); std::printf("\n"); }
}

LIBXSMM_API libxsmm_meqn_function libxsmm_dispatch_meqn(libxsmm_blasint idx, libxsmm_meqn_arg_shape out) {
  if (!rt_ready()) return nullptr;
  std::lock_guard<std::mutex> guard(g_eqn_lock);
  Equation* e = get(idx);
  if (!e || !e->constructed || e->nodes.empty()) return nullptr;      // [ref: libxsmm_matrixeqn.c:1266-1276]
  const std::array<int, 4> key = {out.m, out.n, out.ld, (int)out.type};
  auto hit = e->handles.find(key);
  if (hit != e->handles.end()) return (libxsmm_meqn_function)hit->second;
  if (!infer(*e, 0)) { rt_note("equation refused: shapes/types of the tree cannot be inferred", idx, 0, 0); return nullptr; }
  std::vector<int> order;
  postorder(*e, 0, order);
  EqnPlan* plan = new EqnPlan();
  plan->slot_of.assign(e->nodes.size(), -1);
  size_t max_elems = 1;
  for (int id : order) max_elems = std::max(max_elems, (size_t)e->nodes[id].ld * (size_t)e->nodes[id].n);
  plan->slot_bytes = (max_elems * 8 + 255) & ~(size_t)255;
  // where the result of a node lives: >= 0 an input position (arguments; a ternary GEMM node accumulating into an argument),
  // < 0 the workspace slot -(loc + 1); INT32_MIN: the caller's output (the head)
  std::vector<int> loc(e->nodes.size(), INT32_MIN);
  for (size_t i = 0; i < e->nodes.size(); ++i) if (e->nodes[i].kind == EQ_ARG) loc[i] = e->nodes[i].in_pos;
  for (int id : order) {
    EqnNode nd = e->nodes[id];
    const bool root = (id == 0);
    const GemmNode g = gemm_node(nd.kind, nd.op);
    const bool accumulates = g.is_gemm && nd.kind == EQ_TERNARY;       // result = third operand, in place
    if (root) {   // the head writes the caller's output [ref: matequation ref :28-29; dispatch out shape]
      // the head keeps its own inferred extent; the caller's shape contributes the leading dimension and the type, as in the reference
      // (src/libxsmm_matrixeqn.c:868-936: samples/equation/equation_gather_dot.c declares an M x 1 output for a head that reduces to 1 x 1)
      const bool reducing_head = nd.kind == EQ_UNARY && is_reduce(nd.op);
      // a SCATTER head spreads its operand over the caller's (larger) output: only the leading dimension is the caller's business
      // [ref: generator_matequation_reference_impl.c:41-56: SCATTER exists as the head only, its index list is output.secondary]
      const bool scatter_head = nd.kind == EQ_UNARY && nd.op == LIBXSMM_MELTW_TYPE_UNARY_SCATTER;
      if (((out.m != nd.m || out.n != nd.n) && !(reducing_head && out.m >= nd.m && out.n >= nd.n) && !scatter_head) || (out.ld < nd.m && !scatter_head) || out.ld < 1) {
        rt_note("equation refused: output shape differs from the head node (m, n, ld)", out.m, out.n, out.ld); delete plan; return nullptr;
      }
      nd.ld = out.ld; nd.type = out.type;
      plan->out_scalar = (nd.m == 1 && nd.n == 1); plan->out_scalar_bytes = (size_t)std::max(1, typesize((int)out.type));
    } else if (accumulates) {
      loc[id] = loc[nd.child[2]];
      if (loc[id] < 0 && loc[id] != INT32_MIN) plan->slot_of[id] = -loc[id] - 1;
    } else { plan->slot_of[id] = plan->nslots++; loc[id] = -(plan->slot_of[id] + 1); }
    EqnStep st; std::memset(&st.args, 0, sizeof(st.args));
    st.node = id; st.alpha_from_op = -1; st.dump_from_op = -1; st.idx_from_input = -1; st.root_side = 0; st.scalar_arg[0] = st.scalar_arg[1] = st.scalar_arg[2] = false; st.src[0] = st.src[1] = st.src[2] = INT32_MIN;
    st.gemm = nullptr; st.br_from_op = -1; st.out_loc = loc[id]; st.skip_if_in_place = false;
    MeltwArgs& a = st.args;
    a.nbatch = 1; a.flags = nd.flags; a.type = nd.op; a.comp_type = nd.dtype; a.out_type = nd.type; a.ldo = nd.ld;
    a.in0_type = a.in1_type = a.in2_type = LIBXSMM_DATATYPE_UNSUPPORTED;
    const EqnNode* ch[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < arity(nd.kind); ++c) {
      ch[c] = &e->nodes[nd.child[c]];
      st.src[c] = loc[nd.child[c]];
      st.scalar_arg[c] = ch[c]->kind == EQ_ARG && ch[c]->m == 1 && ch[c]->n == 1;
      if ((ch[c]->kind == EQ_ARG && ch[c]->in_pos < 0) || st.src[c] == INT32_MIN) { delete plan; return nullptr; }
    }
    if (g.is_gemm) {   // one (batch-reduce) GEMM of the dense path: op(A) m x k times op(B) k x n, f32 accumulation
      const EqnNode& A = *ch[0]; const EqnNode& B = *ch[1];
      const int k = g.ta ? A.m : A.n;
      const libxsmm_gemm_shape sh = libxsmm_create_gemm_shape(nd.m, nd.n, k, A.ld, B.ld, nd.ld, (libxsmm_datatype)A.type, (libxsmm_datatype)B.type, (libxsmm_datatype)nd.type, LIBXSMM_DATATYPE_F32);
      const libxsmm_bitfield fl = (g.ta ? LIBXSMM_GEMM_FLAG_TRANS_A : 0) | (g.tb ? LIBXSMM_GEMM_FLAG_TRANS_B : 0) | (g.vnni ? LIBXSMM_GEMM_FLAG_VNNI_A : 0) |
                                  (accumulates ? 0 : LIBXSMM_GEMM_FLAG_BETA_0);
      libxsmm_gemmfunction fn = nullptr;
      if (g.br) {   // the blocks of A and B lie set_stride bytes apart; the count arrives per call in ops_args[op_arg_pos].tertiary
        if (nd.op_arg_pos < 0 || A.kind != EQ_ARG || B.kind != EQ_ARG || A.set == 0 || B.set == 0) { rt_note("equation refused: BRGEMM node without strided argument sets / op argument", nd.op, nd.op_arg_pos, 0); delete plan; return nullptr; }
        fn = libxsmm_dispatch_brgemm(sh, fl, LIBXSMM_GEMM_PREFETCH_NONE, libxsmm_create_gemm_batch_reduce_config(LIBXSMM_GEMM_BATCH_REDUCE_STRIDE, A.set_stride, B.set_stride, 0));
        st.br_from_op = nd.op_arg_pos;
      } else fn = libxsmm_dispatch_gemm(sh, fl, LIBXSMM_GEMM_PREFETCH_NONE);
      if (!fn) { rt_note("equation refused: no GEMM kernel for a MATMUL / BRGEMM node (a, b, c type)", A.type, B.type, nd.type); delete plan; return nullptr; }
      st.gemm = (const void*)fn; plan->has_gemm = true;
      if (root && accumulates) {
        // the head accumulates into its third operand but the result belongs in the caller's output: the operand is copied (and converted) there
        // first, unless the caller passed the very same matrix as input and output -- the usual C += A * B call -- which is checked per call
        const EqnNode& Cn = *ch[2];
        EqnStep cp = st; cp.gemm = nullptr; cp.br_from_op = -1; cp.skip_if_in_place = true;
        cp.src[0] = st.src[2]; cp.src[1] = cp.src[2] = INT32_MIN; cp.scalar_arg[0] = cp.scalar_arg[1] = cp.scalar_arg[2] = false;
        MeltwArgs& c = cp.args;
        c.operation = LIBXSMM_MELTW_OPERATION_UNARY; c.type = LIBXSMM_MELTW_TYPE_UNARY_IDENTITY; c.flags = 0; c.comp_type = LIBXSMM_DATATYPE_F32;
        c.m = nd.m; c.n = nd.n; c.in0_type = Cn.type; c.ldi = Cn.ld; c.out_type = nd.type; c.ldo = nd.ld;
        libxsmm_descriptor_blob cb;
        const libxsmm_meltw_descriptor* cd = libxsmm_meltw_descriptor_init2(&cb, (libxsmm_datatype)c.in0_type, LIBXSMM_DATATYPE_UNSUPPORTED, LIBXSMM_DATATYPE_UNSUPPORTED, LIBXSMM_DATATYPE_F32,
          (libxsmm_datatype)c.out_type, c.m, c.n, c.ldi, c.ldo, 0, 0, 0, (unsigned short)LIBXSMM_MELTW_TYPE_UNARY_IDENTITY, LIBXSMM_MELTW_OPERATION_UNARY);
        if (!cd || !meltw_supported(*cd)) { rt_note("equation refused: the accumulator of the head GEMM cannot be copied to the output (in type, out type)", Cn.type, nd.type, 0); delete plan; return nullptr; }
        plan->steps.push_back(cp);
      }
      plan->steps.push_back(st);
      continue;
    }
    a.in0_type = ch[0]->type; a.ldi = ch[0]->ld;
    libxsmm_descriptor_blob blob;
    const libxsmm_meltw_descriptor* d = nullptr;
    if (nd.kind == EQ_UNARY) {
      a.operation = LIBXSMM_MELTW_OPERATION_UNARY;
      if (is_reduce(nd.op)) { a.m = ch[0]->m; a.n = ch[0]->n; } else { a.m = nd.m; a.n = nd.n; }     // [ref: matequation ref :121-125]
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU) st.alpha_from_op = nd.op_arg_pos;
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_DUMP) st.dump_from_op = nd.op_arg_pos;       // second destination: ops_args[pos].primary [ref: matequation ref :58-60]
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, LIBXSMM_DATATYPE_UNSUPPORTED, LIBXSMM_DATATYPE_UNSUPPORTED, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, 0, 0, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_UNARY);
      // ops with side channels only make sense as standalone TPPs here
      // side channels through output.secondary exist for the head of the tree only [ref: matequation ref :40-55]: the ReLU bitmask,
      // the byte offset of UNZIP's second half
      if ((nd.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) && nd.op == LIBXSMM_MELTW_TYPE_UNARY_RELU && root) st.root_side = 1;
      // (a byte copy in the operand's element size, like the reference's: an operand of another width than the output would be written past the columns)
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_SCATTER && root && typesize((int)ch[0]->type) == typesize((int)nd.type)) st.root_side = 3;
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_UNZIP && root) st.root_side = 2;
      // a GATHER directly above an argument takes its index list from that argument's secondary slot [ref: samples/equation/equation_gather_reduce.c:150-166]
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_GATHER && ch[0]->kind == EQ_ARG) st.idx_from_input = ch[0]->in_pos;
      if (((nd.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) && st.root_side != 1) || (nd.op == LIBXSMM_MELTW_TYPE_UNARY_GATHER && st.idx_from_input < 0) || (nd.op == LIBXSMM_MELTW_TYPE_UNARY_SCATTER && st.root_side != 3) ||
          (nd.op == LIBXSMM_MELTW_TYPE_UNARY_UNZIP && st.root_side != 2) || (nd.op == LIBXSMM_MELTW_TYPE_UNARY_DUMP && nd.op_arg_pos < 0) || nd.op == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR ||
          nd.op == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV || nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV) d = nullptr;
      // parameterised activations: the reference's equation generators do not apply ops_args to them (its CPU JIT leaves the
      // negative side unscaled), so there is no behaviour to be compatible with: refuse instead of guessing
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU) d = nullptr;
    } else if (nd.kind == EQ_BINARY) {
      a.operation = LIBXSMM_MELTW_OPERATION_BINARY; a.m = nd.m; a.n = nd.n;
      if (nd.op == LIBXSMM_MELTW_TYPE_BINARY_MUL_AND_REDUCE_TO_SCALAR_OP_ADD) { a.m = ch[0]->m; a.n = ch[0]->n; }     // the extent of the operands, not of the (1 x 1) result
      a.in1_type = ch[1]->type; a.ldi1 = ch[1]->ld;
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, (libxsmm_datatype)a.in1_type, LIBXSMM_DATATYPE_UNSUPPORTED, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, a.ldi1, 0, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_BINARY);
      if (nd.op >= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT && nd.op <= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE) d = nullptr;
    } else {
      a.operation = LIBXSMM_MELTW_OPERATION_TERNARY; a.m = nd.m; a.n = nd.n;
      a.in1_type = ch[1]->type; a.ldi1 = ch[1]->ld; a.in2_type = ch[2]->type; a.ldi2 = ch[2]->ld;
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, (libxsmm_datatype)a.in1_type, (libxsmm_datatype)a.in2_type, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, a.ldi1, a.ldi2, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_TERNARY);
      if (nd.op == LIBXSMM_MELTW_TYPE_TERNARY_SELECT) d = nullptr;
    }
    if (!d || !meltw_supported(*d)) {
      rt_note("equation refused: node kind/op/flags not available inside a tree", (int)nd.kind, (int)nd.op, (int)nd.flags);
      rt_note("  operand types in0/in1/in2", a.in0_type, a.in1_type, a.in2_type); rt_note("  compute/out type, descriptor built", (int)nd.dtype, (int)nd.type, d != nullptr);
      delete plan; return nullptr;
    }
    // the broadcast flags of an op refer to operands that really are vectors / scalars of the result
    plan->steps.push_back(st);
  }
  if (rt_jit_mode() != 0) {     // element-wise trees: one generated kernel instead of one launch per node
    std::string src, fname; long long total = 0;
    EqnPlan probe;
    if (generate_fused(*e, out, idx, src, fname, probe, total)) {
      std::string why;
      plan->fused = jit_compile(src, fname, total, 16, &why);
      if (!plan->fused && std::getenv("LIBXSMM_HIP_JIT_VERBOSE")) std::fprintf(stderr, "libxsmm_amd: generated equation kernel did not compile: %s\n%s\n", why.c_str(), src.c_str());
      if (plan->fused) { plan->fused_inputs = probe.fused_inputs; plan->fused_scalar = probe.fused_scalar; plan->fused_alphas = probe.fused_alphas; plan->fused_dumps = probe.fused_dumps; }
    }
  }
  const void* h = rt_new_meqn_handle(plan);
  if (!h) { delete plan; return nullptr; }
  e->handles.emplace(key, h);
  return (libxsmm_meqn_function)h;
}

}  // extern "C"
