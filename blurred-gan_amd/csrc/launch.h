// Launch wrapper + step-program recorder (include/bgan.h "step programs": bg_program_*, bg_dstep, bg_gstep).
//
// Every kernel launch of the library goes through bg::launch().  Normally that IS hipLaunchKernelGGL.  While a thread records a
// program (bg_program_record_begin), the launch is turned into a NODE -- kernel address, grid, block, LDS bytes and a by-value copy
// of every kernel argument (the GatherParams-class blocks included) -- which is appended to the program and launched from that very
// copy (so the recording step already runs what a replay runs).  bg_program_replay() walks the nodes and calls hipLaunchKernel on
// each: no geometry checks, no tap tables, no grid planning, no Python.  Per-step scalars (Adam's lr_t, the RNG counter offsets)
// are BOUND: the entry point that owns such an argument asks take_bind() whether the host announced a slot for it and, if so, the
// node re-reads the slot before every replay.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <tuple>
#include <utility>

namespace bg {

struct Node {
  virtual ~Node() {}
  // kind 0: kernel launch, 1: profiling bracket begin, 2: end, 3: executed-flops note, 4: useful-flops note
  int kind = 0;
  virtual hipError_t run(hipStream_t) { return hipSuccess; }
  virtual void* arg_ptr(int) { return nullptr; }
  virtual int n_args() const { return 0; }
  virtual size_t arg_size(int) const { return 0; }
  virtual const void* func() const { return nullptr; }
  virtual void geometry(dim3*, dim3*, size_t*) const {}
  virtual void** argv_ptr() { return nullptr; }
};

template <class... P>
struct LaunchNode final : Node {
  const void* fn;
  dim3 grid, block;
  size_t lds;
  std::tuple<P...> args;
  void* argv[sizeof...(P) ? sizeof...(P) : 1];
  size_t sizes[sizeof...(P) ? sizeof...(P) : 1];

  template <size_t... I>
  void fill(std::index_sequence<I...>) {
    ((argv[I] = static_cast<void*>(&std::get<I>(args))), ...);
    ((sizes[I] = sizeof(std::tuple_element_t<I, std::tuple<P...>>)), ...);
  }
  template <class... A>
  LaunchNode(void (*k)(P...), dim3 g, dim3 b, size_t l, A&&... a)
      : fn(reinterpret_cast<const void*>(k)), grid(g), block(b), lds(l), args(static_cast<P>(std::forward<A>(a))...) {
    kind = 0;
    fill(std::index_sequence_for<P...>{});
  }
  hipError_t run(hipStream_t s) override { return hipLaunchKernel(fn, grid, block, argv, lds, s); }
  void* arg_ptr(int i) override { return (i >= 0 && i < (int)sizeof...(P)) ? argv[i] : nullptr; }
  int n_args() const override { return (int)sizeof...(P); }
  size_t arg_size(int i) const override { return (i >= 0 && i < (int)sizeof...(P)) ? sizes[i] : 0; }
  const void* func() const override { return fn; }
  void geometry(dim3* g, dim3* b, size_t* l) const override { *g = grid; *b = block; *l = lds; }
  void** argv_ptr() override { return argv; }
};

// ---- recorder state (program.hip); thread-local: one thread records one program at a time
bool recording();
void rec_push(Node* n, hipStream_t s);            // appends and launches a kernel node (errors surface through hipGetLastError)
void rec_note(int kind, const char* name, double a, double b);     // profiling bracket nodes

enum BindKind { BIND_F32_FROM_F64 = 0, BIND_U64 = 1 };
// The host announced (bg_program_bind_next) that argument `what` of the next launch reads a slot: returns the slot or -1.
int take_bind(int what);
// Marks argument `arg_index` of the node recorded LAST as bound to `slot` (no-op when slot < 0 or nothing is being recorded).
void bind_last(int arg_index, BindKind kind, int slot);

template <class... P, class... A>
inline void launch(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "bg::launch: argument count differs from the kernel's parameter list");
  if (!recording()) {
    hipLaunchKernelGGL(kern, grid, block, lds, s, static_cast<P>(std::forward<A>(a))...);
    return;
  }
  rec_push(new LaunchNode<P...>(kern, grid, block, lds, std::forward<A>(a)...), s);
}

}  // namespace bg
