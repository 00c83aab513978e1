// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  A minimal CPU stand-in for the handful of HIP device
// constructs used by highwayenv_amd/csrc/hwy_device.h and hwy_wave.h, so that the *same kernel
// source* can be executed on the CPU and checked against the golden traces in the build container,
// which has no GPU.  Never compiled into, linked with or loaded by the product.
//
// Execution model: every GPU thread of a workgroup is a FIBER (ucontext) on one OS thread,
// scheduled round-robin; __syncthreads / ballots / readlane / ds_permute are rendezvous points.
// Workgroups run one after the other.  Restriction honoured by the kernels: every rendezvous is
// executed in workgroup-uniform control flow.
#pragma once
#include <math.h>
#include <stdint.h>
#include <ucontext.h>

#include <cstdlib>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline
#define HWY_FMA_K(a, b, c) fma((a), (b), (c))
// hwy_math.h, the paired forms (two evaluations sharing the coefficient): the same operations, one after the other
#define HWY_FMA_K2(r0, r1, a0, b0, a1, b1, c) do { const double t0_ = fma((a0), (b0), (c)), t1_ = fma((a1), (b1), (c)); (r0) = t0_; (r1) = t1_; } while (0)
#define HWY_LEAD2(r0, r1, t0, t1, kn, kn1) do { (r0) = ::hwy::lead_unfused((t0), (kn), (kn1)); (r1) = ::hwy::lead_unfused((t1), (kn), (kn1)); } while (0)
#define HWY_RELOAD_PARAMS(q, p) const StepParams &q = p  // hwy_wave.h: re-read of the kernel-argument segment
#define HWY_RELOAD_STEP_PARAMS(q, p) const StepParams &q = p  // hwy_device.h
#define HWY_RELOAD_IX_PARAMS(q, ip) const IxParams &q = ip  // hwy_ix.h
#define HWY_RELOAD_NET_PARAMS(q, np) const NetParams &q = np  // hwy_net.h: the same for the road-network kernels
#define HWY_PIN_POINTERS(a, b) ((void)0)              // hwy_device.h: store_vehicle_at
#define HWY_GLOBAL_F64 double
#define HWY_KERNARG_TOUCH(T) ((void)0)                // hwy_wave.h: a prefetch of the kernel-argument segment (device build only)
#define HWY_ISSUED_TOGETHER(a, b, c, d, e_) ((void)0)  // hwy_wave.h: a scheduling constraint of the device build only
#define HWY_WAVE_LDS_FENCE() __syncthreads()  // hwy_wave.h: the 64 fibers of a workgroup need a real rendezvous
#define HWY_SAT_FENCE() ((void)0)                     // hwy_device.h: scheduling / register-allocation constraints of the SAT (device build only)
#define HWY_SAT_SETTLE(f) ((void)0)
#define HWY_WAVEFRONT_FENCE() emu::wave_barrier()          // hwy_device.h: LDS handed over within one wavefront of a workgroup
#define HWY_WAVE_MAX_U32(v) emu::wave_max_u32(v)          // hwy_device.h: DPP reduction on the device
#define HWY_KC(c) (c)  // hwy_math.h: SGPR-pinned constant (an AMDGPU inline-asm constraint on the device)

struct emu_dim3 { int x = 0, y = 0, z = 0; };

namespace emu {
struct Fiber {
  ucontext_t ctx;
  emu_dim3 tid, bid, bdim;
  unsigned ballot_phase = 0;
  unsigned xchg_phase = 0;
  unsigned bor_phase = 0;
  bool finished = false;
  char *stack = nullptr;
};
inline Fiber *cur = nullptr;
inline std::vector<Fiber> *fibers = nullptr;
inline ucontext_t main_ctx;
inline int n_fibers = 0;
// generation barrier
inline int bar_count = 0;
inline unsigned bar_gen = 0;

inline void yield() {
  Fiber *me = cur;
  const int next = (me->tid.x + 1) % n_fibers;
  cur = &(*fibers)[next];
  swapcontext(&me->ctx, &cur->ctx);
}
inline void barrier() {
  const unsigned gen = bar_gen;
  if (++bar_count == n_fibers) {
    bar_count = 0;
    ++bar_gen;
  } else {
    while (bar_gen == gen) yield();
  }
}
// rendezvous of the (up to) 64 fibers of ONE wavefront: ballots / readlane / ds_permute are wave-level operations, and the
// wavefronts of a multi-wave workgroup may execute different numbers of them between two workgroup barriers
inline int wbar_count[16];
inline unsigned wbar_gen[16];
inline void wave_barrier() {
  const int w = cur->tid.x >> 6;
  const int size = n_fibers - 64 * w < 64 ? n_fibers - 64 * w : 64;
  const unsigned gen = wbar_gen[w];
  if (++wbar_count[w] == size) {
    wbar_count[w] = 0;
    ++wbar_gen[w];
  } else {
    while (wbar_gen[w] == gen) yield();
  }
}
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)

inline void __syncthreads() { emu::barrier(); }
inline void __threadfence() {}  // fibers of one OS thread: program order is memory order
inline void __threadfence_block() {}

namespace emu {
inline unsigned long long g_ballot[2][16];
inline int g_xchg[2][1024];
}  // namespace emu

inline unsigned long long __ballot(int pred) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long &acc = emu::g_ballot[emu::cur->ballot_phase & 1][wave];
  emu::cur->ballot_phase++;
  if (pred) acc |= 1ull << lane;
  emu::wave_barrier();
  const unsigned long long v = acc;
  emu::wave_barrier();
  if (lane == 0) acc = 0;  // reused two ballots later, with >= 1 rendezvous in between
  return v;
}
inline int __syncthreads_or(int pred) {  // barrier + OR of the predicate over the whole workgroup
  unsigned long long any = __ballot(pred);  // per-wave OR (two wave rendezvous) ...
  __shared__ unsigned long long acc[2];
  const unsigned ph = emu::cur->bor_phase++ & 1;  // (its own counter: the wavefronts' ballot counts may differ)
  if (any) acc[ph] = 1;
  emu::barrier();
  const int v = acc[ph] != 0;
  emu::barrier();
  if (threadIdx.x == 0) acc[ph] = 0;
  return v;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// ---- wave data movement (hwy_wave.h; 64-thread workgroups, workgroup-uniform calls) --------------
namespace emu {
// double-buffered exchange: one rendezvous per call (a fiber can be at most one call ahead)
inline int readlane(int v, int lane) {
  int *buf = g_xchg[cur->xchg_phase++ & 1];
  buf[threadIdx.x] = v;
  wave_barrier();
  return buf[(threadIdx.x & ~63) + (lane & 63)];
}
inline int ds_permute(int addr, int v) {  // my value lands in lane (addr/4)%64 (a permutation in our use)
  int *buf = g_xchg[cur->xchg_phase++ & 1];
  buf[(threadIdx.x & ~63) + ((addr >> 2) & 63)] = v;
  wave_barrier();
  return buf[threadIdx.x];
}
inline int ds_bpermute(int addr, int v) {  // I read the value lane (addr/4)%64 holds
  int *buf = g_xchg[cur->xchg_phase++ & 1];
  buf[threadIdx.x] = v;
  wave_barrier();
  return buf[(threadIdx.x & ~63) + ((addr >> 2) & 63)];
}
}  // namespace emu
namespace emu {
inline unsigned wave_max_u32(unsigned v) {  // butterfly over the 64 fibers of the wavefront
  for (int st = 1; st < 64; st <<= 1) {
    const unsigned o = (unsigned)ds_bpermute((((int)threadIdx.x & 63) ^ st) << 2, (int)v);
    v = o > v ? o : v;
  }
  return v;
}
}  // namespace emu
#define __builtin_amdgcn_ds_bpermute(addr, v) emu::ds_bpermute((addr), (v))
#define __builtin_amdgcn_readlane(v, lane) emu::readlane((v), (lane))
#define __builtin_amdgcn_readfirstlane(v) (v)  // (only used on wave-uniform values: every fiber holds the same one)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt(x))
#define __builtin_amdgcn_ds_permute(addr, v) emu::ds_permute((addr), (v))
// agent-scope atomics (fibers run on one OS thread: plain accesses)
template <typename T, typename U> inline T atomicAdd(T *p, U v) { const T old = *p; *p = old + (T)v; return old; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_store(ptr, v, order, scope) (*(ptr) = (v))
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
#define __hip_atomic_fetch_min(ptr, v, order, scope) (*(ptr) = (*(ptr) < (v) ? *(ptr) : (v)))
#define __hip_atomic_fetch_or(ptr, v, order, scope) (*(ptr) |= (v))
#define __hip_atomic_fetch_max(ptr, v, order, scope) (*(ptr) = (*(ptr) > (v) ? *(ptr) : (v)))
inline double __longlong_as_double(long long v) { double d; __builtin_memcpy(&d, &v, 8); return d; }
inline long long __double_as_longlong(double d) { long long v; __builtin_memcpy(&v, &d, 8); return v; }
inline int __double2loint(double d) { long long b; __builtin_memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
inline int __double2hiint(double d) { long long b; __builtin_memcpy(&b, &d, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo) {
  const long long b = ((long long)hi << 32) | (unsigned int)lo;
  double d; __builtin_memcpy(&d, &b, 8); return d;
}

namespace emu {
inline std::function<void()> g_body;
inline int g_grid = 0;
inline int g_done = 0;
inline void fiber_main() {
  for (int b = 0; b < g_grid; ++b) {
    cur->bid.x = b;
    g_body();
    barrier();  // the whole workgroup finishes a block before the next one starts
  }
  cur->finished = true;
  if (++g_done == n_fibers) {
    setcontext(&main_ctx);  // last fiber: back to launch()
  }
  for (;;) yield();
}
// run `kernel(params)` for `grid` workgroups of `block` threads
template <typename K, typename P>
void launch(K kernel, int grid, int block, const P &params) {
  constexpr size_t kStack = 512 * 1024;
  std::vector<Fiber> fs(block);
  fibers = &fs;
  n_fibers = block;
  bar_count = 0;
  bar_gen = 0;
  for (int w = 0; w < 16; ++w) { wbar_count[w] = 0; wbar_gen[w] = 0; }
  g_done = 0;
  g_grid = grid;
  for (auto &row : g_ballot)
    for (auto &a : row) a = 0;
  g_body = [&] { kernel(params); };
  for (int t = 0; t < block; ++t) {
    Fiber &f = fs[t];
    f.tid.x = t;
    f.bdim.x = block;
    f.stack = (char *)std::malloc(kStack);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_main, 0);
  }
  cur = &fs[0];
  swapcontext(&main_ctx, &fs[0].ctx);
  for (auto &f : fs) std::free(f.stack);
  fibers = nullptr;
  cur = nullptr;
}
}  // namespace emu
