// Step programs: record a step's launch list once, replay it with one call (include/bgan.h bg_program_*, bg_dstep, bg_gstep).
// Replaces the per-batch Python loop over TF ops of the reference's train_on_batch (wgan.py:86-114,132-172) on the host side:
// the kernels and their arguments are exactly those the eager path issued on the recording step.
#include "common.h"
#include <map>
#include <memory>
#include <string>
#include <vector>

struct bg_program {
  struct Bind {
    int node, arg, kind, slot;
  };
  struct NoteNode final : bg::Node {
    std::string name;
    double a = 0, b = 0;
  };
  struct GraphForm {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> knodes;        // one per kernel node of the range, in order
    std::vector<int> node_index;               // program node index of each graph kernel node
  };
  std::vector<std::unique_ptr<bg::Node>> nodes;
  std::vector<Bind> binds;
  std::vector<double> f64;
  std::vector<uint64_t> u64;
  std::map<std::pair<int, int>, GraphForm> graphs;
  int launches = 0;
  bool recording = false;
  hipStream_t graph_stream = nullptr;     // the stream the last bg_program_graph_launch went to
  bool graph_launched = false;

  ~bg_program() { drop_graphs(); }
  void drop_graphs() {
    // an executable graph may still be running (eviction right after a launch): wait for the stream it was launched on -- not for
    // the device; the library promises no hidden device-wide synchronisation (include/bgan.h "Threading / async")
    if (!graphs.empty() && graph_launched) (void)hipStreamSynchronize(graph_stream);
    graph_launched = false;
    for (auto& kv : graphs) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    graphs.clear();
  }
  // writes the current slot values into the argument copies of the bound nodes in [first, last)
  void apply_binds(int first, int last) {
    for (const Bind& b : binds) {
      if (b.node < first || b.node >= last) continue;
      void* dst = nodes[b.node]->arg_ptr(b.arg);
      if (b.kind == bg::BIND_F32_FROM_F64) *static_cast<float*>(dst) = (float)f64[b.slot];
      else *static_cast<uint64_t*>(dst) = u64[b.slot];
    }
  }
};

namespace bg {

namespace {
thread_local bg_program* t_rec = nullptr;
thread_local int t_pending_what[4] = {0, 0, 0, 0};
thread_local int t_pending_slot[4] = {-1, -1, -1, -1};
thread_local bool t_bind_failed = false;       // a bind_last() that did not fit its node: reported by bg_program_record_end
}  // namespace

bool recording() { return t_rec != nullptr; }

void rec_push(Node* n, hipStream_t s) {
  t_rec->nodes.emplace_back(n);
  t_rec->launches++;
  (void)n->run(s);          // a failure stays in the runtime's last-error slot, where Launch::done() reads it
}

void rec_note(int kind, const char* name, double a, double b) {
  if (!t_rec) return;
  auto* n = new bg_program::NoteNode();
  n->kind = kind;
  if (name) n->name = name;
  n->a = a;
  n->b = b;
  t_rec->nodes.emplace_back(n);
}

int take_bind(int what) {
  if (!t_rec) return -1;
  for (int i = 0; i < 4; ++i)
    if (t_pending_slot[i] >= 0 && t_pending_what[i] == what) {
      const int s = t_pending_slot[i];
      t_pending_slot[i] = -1;
      return s;
    }
  return -1;
}

void bind_last(int arg_index, BindKind kind, int slot) {
  if (!t_rec || slot < 0 || t_rec->nodes.empty()) return;
  const int node = (int)t_rec->nodes.size() - 1;
  Node* n = t_rec->nodes[node].get();
  const size_t want = kind == BIND_F32_FROM_F64 ? sizeof(float) : sizeof(uint64_t);
  if (n->kind != 0 || n->arg_size(arg_index) != want) {
    // a silently dropped binding would replay the recording step's lr_t / Philox offset for ever: fail the recording instead
    set_error("step program: binding of slot %d to argument %d of node %d does not fit (node kind %d, argument size %zu, wanted %zu)",
              slot, arg_index, node, n->kind, n->arg_size(arg_index), want);
    t_bind_failed = true;
    return;
  }
  t_rec->binds.push_back({node, arg_index, (int)kind, slot});
}

}  // namespace bg

extern "C" {

int bg_program_create(bg_program** out, int n_slots) {
  BG_REQUIRE(out, BG_ERR_NULL, "bg_program_create: null out pointer");
  BG_REQUIRE(n_slots >= 0 && n_slots <= 4096, BG_ERR_BAD_SHAPE, "bg_program_create: n_slots=%d", n_slots);
  auto* p = new bg_program();
  p->f64.assign((size_t)n_slots + 1, 0.0);
  p->u64.assign((size_t)n_slots + 1, 0);
  *out = p;
  return BG_OK;
}

int bg_program_destroy(bg_program* p) {
  if (p && bg::t_rec == p) bg::t_rec = nullptr;
  delete p;
  return BG_OK;
}

int bg_program_record_begin(bg_program* p) {
  BG_REQUIRE(p, BG_ERR_NULL, "bg_program_record_begin: null program");
  BG_REQUIRE(!bg::t_rec, BG_ERR_UNSUPPORTED, "bg_program_record_begin: this thread is already recording a program");
  p->drop_graphs();
  p->nodes.clear();
  p->binds.clear();
  p->launches = 0;
  p->recording = true;
  for (int i = 0; i < 4; ++i) bg::t_pending_slot[i] = -1;
  bg::t_bind_failed = false;
  bg::t_rec = p;
  return BG_OK;
}

int bg_program_record_end(bg_program* p) {
  BG_REQUIRE(p, BG_ERR_NULL, "bg_program_record_end: null program");
  BG_REQUIRE(bg::t_rec == p, BG_ERR_UNSUPPORTED, "bg_program_record_end: this program is not the one being recorded");
  bg::t_rec = nullptr;
  p->recording = false;
  if (bg::t_bind_failed) {
    bg::t_bind_failed = false;
    return BG_ERR_UNSUPPORTED;          // message left by bind_last
  }
  for (int i = 0; i < 4; ++i)
    BG_REQUIRE(bg::t_pending_slot[i] < 0, BG_ERR_UNSUPPORTED,
               "bg_program_record_end: a binding (what=%d) was announced but no launch consumed it", bg::t_pending_what[i]);
  return BG_OK;
}

int bg_program_size(const bg_program* p) { return p ? (int)p->nodes.size() : 0; }

int bg_program_launches(const bg_program* p) { return p ? p->launches : 0; }

int bg_program_bind_next(int what, int slot) {
  BG_REQUIRE(bg::t_rec, BG_ERR_UNSUPPORTED, "bg_program_bind_next: no program is being recorded on this thread");
  BG_REQUIRE(what == BG_BIND_ADAM_LR || what == BG_BIND_RNG_OFFSET, BG_ERR_BAD_SHAPE, "bg_program_bind_next: what=%d", what);
  BG_REQUIRE(slot >= 0 && (size_t)slot + 1 < bg::t_rec->f64.size() + 0, BG_ERR_BAD_SHAPE, "bg_program_bind_next: slot %d out of range", slot);
  for (int i = 0; i < 4; ++i)
    if (bg::t_pending_slot[i] < 0) {
      bg::t_pending_what[i] = what;
      bg::t_pending_slot[i] = slot;
      return BG_OK;
    }
  return bg::fail(BG_ERR_UNSUPPORTED, "bg_program_bind_next: too many bindings waiting for a launch");
}

int bg_program_binds(const bg_program* p) { return p ? (int)p->binds.size() : 0; }

double* bg_program_slots_f64(bg_program* p) { return p ? p->f64.data() : nullptr; }

uint64_t* bg_program_slots_u64(bg_program* p) { return p ? p->u64.data() : nullptr; }

int bg_program_replay(bg_program* p, int first, int last, void* stream) {
  BG_REQUIRE(p, BG_ERR_NULL, "bg_program_replay: null program");
  BG_REQUIRE(!p->recording, BG_ERR_UNSUPPORTED, "bg_program_replay: the program is still being recorded");
  const int n = (int)p->nodes.size();
  if (last < 0 || last > n) last = n;
  BG_REQUIRE(first >= 0 && first <= last, BG_ERR_BAD_SHAPE, "bg_program_replay: range [%d, %d) of %d nodes", first, last, n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  p->apply_binds(first, last);
  const bool prof = bg::prof_on();
  for (int i = first; i < last; ++i) {
    bg::Node* nd = p->nodes[i].get();
    if (nd->kind == 0) {
      hipError_t e = nd->run(s);
      if (e != hipSuccess) return bg::fail(BG_ERR_HIP, "bg_program_replay: launch %d: %s", i, hipGetErrorString(e));
    } else if (prof) {
      auto* nn = static_cast<bg_program::NoteNode*>(nd);
      if (nd->kind == 1) bg::prof_begin(s, nn->name.c_str(), nn->a, nn->b);
      else if (nd->kind == 2) bg::prof_end(s);
      else if (nd->kind == 3) bg::prof_exec_flops(nn->a);
      else if (nd->kind == 4) bg::prof_useful_flops(nn->a);
    }
  }
  return BG_OK;
}

int bg_dstep(bg_program* p, void* stream) { return bg_program_replay(p, 0, -1, stream); }

int bg_gstep(bg_program* p, void* stream) { return bg_program_replay(p, 0, -1, stream); }

// The same node range as ONE hipGraph launch: kernel nodes in a linear chain (the stream order of the recording), built on first
// use; bound arguments are refreshed with hipGraphExecKernelNodeSetParams before each launch.  Profiling brackets are not part of
// a graph: with bg_prof_enable(1) the call falls back to the node-by-node replay.
int bg_program_graph_launch(bg_program* p, int first, int last, void* stream) {
  BG_REQUIRE(p, BG_ERR_NULL, "bg_program_graph_launch: null program");
  BG_REQUIRE(!p->recording, BG_ERR_UNSUPPORTED, "bg_program_graph_launch: the program is still being recorded");
  if (bg::prof_on()) return bg_program_replay(p, first, last, stream);
  const int n = (int)p->nodes.size();
  if (last < 0 || last > n) last = n;
  BG_REQUIRE(first >= 0 && first <= last, BG_ERR_BAD_SHAPE, "bg_program_graph_launch: range [%d, %d) of %d nodes", first, last, n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  p->apply_binds(first, last);
  auto key = std::make_pair(first, last);
  auto it = p->graphs.find(key);
#define BG_HIP(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) return bg::fail(BG_ERR_HIP, "bg_program_graph_launch: %s: %s", #call, hipGetErrorString(e_)); \
  } while (0)
  auto params_of = [](bg::Node* nd) {
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof kp);
    dim3 g, b;
    size_t lds = 0;
    nd->geometry(&g, &b, &lds);
    kp.func = const_cast<void*>(nd->func());
    kp.gridDim = g;
    kp.blockDim = b;
    kp.sharedMemBytes = (unsigned)lds;
    kp.kernelParams = nd->argv_ptr();
    kp.extra = nullptr;
    return kp;
  };
  if (it == p->graphs.end()) {
    bg_program::GraphForm gf;
    BG_HIP(hipGraphCreate(&gf.graph, 0));
    hipGraphNode_t prev = nullptr;
    for (int i = first; i < last; ++i) {
      bg::Node* nd = p->nodes[i].get();
      if (nd->kind != 0) continue;
      hipKernelNodeParams kp = params_of(nd);
      hipGraphNode_t gn;
      BG_HIP(hipGraphAddKernelNode(&gn, gf.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
      gf.knodes.push_back(gn);
      gf.node_index.push_back(i);
      prev = gn;
    }
    BG_HIP(hipGraphInstantiate(&gf.exec, gf.graph, nullptr, nullptr, 0));
    it = p->graphs.emplace(key, std::move(gf)).first;
  } else {
    bg_program::GraphForm& gf = it->second;
    for (const bg_program::Bind& b : p->binds) {
      if (b.node < first || b.node >= last) continue;
      for (size_t k = 0; k < gf.node_index.size(); ++k)
        if (gf.node_index[k] == b.node) {
          hipKernelNodeParams kp = params_of(p->nodes[b.node].get());
          BG_HIP(hipGraphExecKernelNodeSetParams(gf.exec, gf.knodes[k], &kp));
          break;
        }
    }
  }
  BG_HIP(hipGraphLaunch(it->second.exec, s));
  p->graph_stream = s;
  p->graph_launched = true;
#undef BG_HIP
  return BG_OK;
}

}  // extern "C"
