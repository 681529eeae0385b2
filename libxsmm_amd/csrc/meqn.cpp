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
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "internal.hpp"

namespace xamd {

enum { EQ_NONE = 0, EQ_ARG = 1, EQ_UNARY = 2, EQ_BINARY = 3, EQ_TERNARY = 4 };

struct EqnNode {
  int kind = EQ_NONE;
  int op = 0, dtype = 0; unsigned int flags = 0; int op_arg_pos = -1;   // op nodes
  int in_pos = -1, set = 0;                                             // arg nodes
  int child[3] = {-1, -1, -1}, up = -1;
  int m = 0, n = 0, ld = 0, type = 0;                                   // result shape / datatype of this node
};

struct Equation {
  std::vector<EqnNode> nodes;       // nodes[0] is the root
  int cur = 0;                      // node that takes the next push
  bool constructed = false;
  std::map<std::array<int, 4>, const void*> handles;   // dispatched (m, n, ld, type) -> handle
};

struct EqnStep { MeltwArgs args; int src[3]; int node; int alpha_from_op; };   // src: >=0 input position, < 0: -(slot+1)
struct EqnPlan {
  std::vector<EqnStep> steps;
  std::vector<int> slot_of;         // per node: workspace slot (-1: none)
  size_t slot_bytes = 0; int nslots = 0;
};

namespace {

std::mutex g_eqn_lock;
std::vector<Equation*> g_eqns;

int arity(int kind) { return kind == EQ_UNARY ? 1 : kind == EQ_BINARY ? 2 : kind == EQ_TERNARY ? 3 : 0; }

bool is_reduce(int t) {
  return t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ADD || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MAX || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_ABSMAX ||
         t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MIN || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X_OP_MUL || t == LIBXSMM_MELTW_TYPE_UNARY_REDUCE_X2_OP_ADD;
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
  if (nd.kind == EQ_ARG) return nd.m > 0 && nd.n > 0 && nd.ld >= nd.m && nd.set == 0;
  for (int c = 0; c < arity(nd.kind); ++c) if (nd.child[c] < 0 || !infer(e, nd.child[c])) return false;
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

}  // namespace

void free_meqn_plan(EqnPlan* plan) { delete plan; }

void run_meqn(EqnPlan* plan, const void* param) {
  const libxsmm_meqn_param* p = (const libxsmm_meqn_param*)param;
  if (!p->inputs || !p->output.primary) { set_error(-2, "matrix equation called without inputs / output"); return; }
  char* ws = nullptr;
  if (plan->nslots > 0) { ws = (char*)rt_workspace(plan->slot_bytes * (size_t)plan->nslots); if (!ws) return; }
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
    }
    a.in0 = src[0]; a.in1 = src[1]; a.in2 = src[2];
    a.out = (s + 1 == plan->steps.size()) ? (char*)p->output.primary : ws + plan->slot_bytes * (size_t)plan->slot_of[st.node];
    if (st.alpha_from_op >= 0) {
      if (!p->ops_args || !p->ops_args[st.alpha_from_op].primary) { set_error(-2, "matrix equation: op argument %d is NULL", st.alpha_from_op); return; }
      a.scalar_f32 = *(const float*)p->ops_args[st.alpha_from_op].primary;
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
  nd.set = (attr.type == LIBXSMM_MATRIX_ARG_TYPE_SET) ? 1 : 0;
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
  if (!infer(*e, 0)) return nullptr;
  std::vector<int> order;
  postorder(*e, 0, order);
  EqnPlan* plan = new EqnPlan();
  plan->slot_of.assign(e->nodes.size(), -1);
  size_t max_elems = 1;
  for (int id : order) max_elems = std::max(max_elems, (size_t)e->nodes[id].ld * (size_t)e->nodes[id].n);
  plan->slot_bytes = (max_elems * 8 + 255) & ~(size_t)255;
  for (int id : order) {
    EqnNode nd = e->nodes[id];
    const bool root = (id == 0);
    if (root) {   // the head writes the caller's output [ref: matequation ref :28-29; dispatch out shape]
      if (out.m != nd.m || out.n != nd.n || out.ld < out.m) { delete plan; return nullptr; }
      nd.ld = out.ld; nd.type = out.type;
    } else plan->slot_of[id] = plan->nslots++;
    EqnStep st; std::memset(&st.args, 0, sizeof(st.args));
    st.node = id; st.alpha_from_op = -1; st.src[0] = st.src[1] = st.src[2] = INT32_MIN;
    MeltwArgs& a = st.args;
    a.nbatch = 1; a.flags = nd.flags; a.type = nd.op; a.comp_type = nd.dtype; a.out_type = nd.type; a.ldo = nd.ld;
    a.in0_type = a.in1_type = a.in2_type = LIBXSMM_DATATYPE_UNSUPPORTED;
    const EqnNode* ch[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < arity(nd.kind); ++c) {
      ch[c] = &e->nodes[nd.child[c]];
      st.src[c] = ch[c]->kind == EQ_ARG ? ch[c]->in_pos : -(plan->slot_of[nd.child[c]] + 1);
      if (ch[c]->kind == EQ_ARG && ch[c]->in_pos < 0) { delete plan; return nullptr; }
    }
    a.in0_type = ch[0]->type; a.ldi = ch[0]->ld;
    libxsmm_descriptor_blob blob;
    const libxsmm_meltw_descriptor* d = nullptr;
    if (nd.kind == EQ_UNARY) {
      a.operation = LIBXSMM_MELTW_OPERATION_UNARY;
      if (is_reduce(nd.op)) { a.m = ch[0]->m; a.n = ch[0]->n; } else { a.m = nd.m; a.n = nd.n; }     // [ref: matequation ref :121-125]
      if (nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU) st.alpha_from_op = nd.op_arg_pos;
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, LIBXSMM_DATATYPE_UNSUPPORTED, LIBXSMM_DATATYPE_UNSUPPORTED, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, 0, 0, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_UNARY);
      // ops with side channels only make sense as standalone TPPs here
      if ((nd.flags & LIBXSMM_MELTW_FLAG_UNARY_BITMASK_2BYTEMULT) || nd.op == LIBXSMM_MELTW_TYPE_UNARY_GATHER || nd.op == LIBXSMM_MELTW_TYPE_UNARY_SCATTER ||
          nd.op == LIBXSMM_MELTW_TYPE_UNARY_UNZIP || nd.op == LIBXSMM_MELTW_TYPE_UNARY_DUMP || nd.op == LIBXSMM_MELTW_TYPE_UNARY_REPLICATE_COL_VAR ||
          nd.op == LIBXSMM_MELTW_TYPE_UNARY_RELU_INV || nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU_INV || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU_INV) d = nullptr;
      if (st.alpha_from_op < 0 && (nd.op == LIBXSMM_MELTW_TYPE_UNARY_LEAKY_RELU || nd.op == LIBXSMM_MELTW_TYPE_UNARY_ELU)) d = nullptr;
    } else if (nd.kind == EQ_BINARY) {
      a.operation = LIBXSMM_MELTW_OPERATION_BINARY; a.m = nd.m; a.n = nd.n;
      a.in1_type = ch[1]->type; a.ldi1 = ch[1]->ld;
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, (libxsmm_datatype)a.in1_type, LIBXSMM_DATATYPE_UNSUPPORTED, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, a.ldi1, 0, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_BINARY);
      if (nd.op == LIBXSMM_MELTW_TYPE_BINARY_ZIP || (nd.op >= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_GT && nd.op <= LIBXSMM_MELTW_TYPE_BINARY_CMP_OP_NE)) d = nullptr;
    } else {
      a.operation = LIBXSMM_MELTW_OPERATION_TERNARY; a.m = nd.m; a.n = nd.n;
      a.in1_type = ch[1]->type; a.ldi1 = ch[1]->ld; a.in2_type = ch[2]->type; a.ldi2 = ch[2]->ld;
      d = libxsmm_meltw_descriptor_init2(&blob, (libxsmm_datatype)a.in0_type, (libxsmm_datatype)a.in1_type, (libxsmm_datatype)a.in2_type, (libxsmm_datatype)nd.dtype,
        (libxsmm_datatype)nd.type, a.m, a.n, a.ldi, a.ldo, a.ldi1, a.ldi2, (unsigned short)nd.flags, (unsigned short)nd.op, LIBXSMM_MELTW_OPERATION_TERNARY);
      if (nd.op == LIBXSMM_MELTW_TYPE_TERNARY_SELECT) d = nullptr;
    }
    if (!d || !meltw_supported(*d)) { delete plan; return nullptr; }
    // the broadcast flags of an op refer to operands that really are vectors / scalars of the result
    plan->steps.push_back(st);
  }
  const void* h = rt_new_meqn_handle(plan);
  if (!h) { delete plan; return nullptr; }
  e->handles.emplace(key, h);
  return (libxsmm_meqn_function)h;
}

}  // extern "C"
