// Developer tool (GPU box): issue cost of the instruction classes the step kernels are made of, on gfx950.
//
// For every class: ONE kernel whose waves run a loop of 64 dependence-free instructions of that class (8 independent register
// chains, rotated) ITERS times; launched with 1024 x W workgroups of one wavefront (W = 1, 2, 4 wavefronts per SIMD on the 1024
// SIMDs of an MI355X).  Reported per class and W: shader-clock ticks (s_memtime) per instruction per WAVE, and the launch
// duration (hipEvents) per instruction per SIMD in ns -- the second one times the measured shader clock is the issue cost in
// cycles the roofline's VALU-issue floor should price that class at (bench.py: valu_view, profiles/r05_issue_costs.json).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/issue_bench tools/microbench/issue_bench.hip
//   tools/microbench/issue_bench > gpurun_out/issue_costs.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2000;
constexpr int PER_ITER = 64;

struct Out { unsigned long long ticks; unsigned long long realtime; double sink; };

// 8 independent chains a0..a7 (f64), i0..i7 (b32); k/k2 loop-invariant operands
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP64(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)

#define KERNEL_HEAD(name)                                                                              \
  __global__ void __launch_bounds__(64) name(Out *out, double seed) {                                  \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double k = 0.999999 + seed * 1e-9, k2 = seed * 1e-7;                                         \
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7; \
    const int ik = (int)seed | 0x55aa;                                                                 \
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7; \
    const float fk = (float)k, fk2 = (float)k2;                                                        \
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;                                \
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;                 \
    (void)fk2; (void)ik; (void)k2;                                                                     \
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();                   \
    for (int it = 0; it < ITERS; ++it) {

#define KERNEL_TAIL                                                                                    \
    }                                                                                                  \
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();                   \
    double sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) +  \
                  (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + (double)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + \
                  (double)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7);                                     \
    if (threadIdx.x == 0) { out[blockIdx.x].ticks = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }     \
    if (sink == 1.2345e-300) out[blockIdx.x].sink = sink;                                              \
  }

// one asm statement per loop body: 8 rounds over 8 independent chains (%0 .. %7), loop-invariant operands %8 / %9 -- nothing
// of the compiler's (hazard s_nops between asm statements, copies) gets between the instructions
#define X8(S) S("0") S("1") S("2") S("3") S("4") S("5") S("6") S("7")
#define X64(S) X8(S) X8(S) X8(S) X8(S) X8(S) X8(S) X8(S) X8(S)
#define X32(S) X8(S) X8(S) X8(S) X8(S)
#define V8(x) "+v"(x##0), "+v"(x##1), "+v"(x##2), "+v"(x##3), "+v"(x##4), "+v"(x##5), "+v"(x##6), "+v"(x##7)
#define S8(x) "+s"(x##0), "+s"(x##1), "+s"(x##2), "+s"(x##3), "+s"(x##4), "+s"(x##5), "+s"(x##6), "+s"(x##7)

#define T_FMA64(n) "v_fma_f64 %" n ", %" n ", %8, %9\n"
#define T_MUL64(n) "v_mul_f64 %" n ", %" n ", %8\n"
#define T_ADD64(n) "v_add_f64 %" n ", %" n ", %9\n"
#define T_MAX64(n) "v_max_f64 %" n ", %" n ", %9\n"
#define T_RCP64(n) "v_rcp_f64 %" n ", %" n "\n"
#define T_RSQ64(n) "v_rsq_f64 %" n ", %" n "\n"
#define T_CMP64(n) "v_cmp_lt_f64 %" n ", %8, %9\n"           /* %0..%7: SGPR pairs */
#define T_CMP32(n) "v_cmp_lt_i32 %" n ", %8, %9\n"
#define T_CNDMASK(n) "v_cndmask_b32 %" n ", %" n ", %8, %9\n" /* %9: an SGPR pair */
#define T_MOV32(n) "v_mov_b32 %" n ", %8\n"
#define T_AND32(n) "v_and_b32 %" n ", %" n ", %8\n"
#define T_ADD32(n) "v_add_u32 %" n ", %" n ", %8\n"
#define T_LSHL64(n) "v_lshlrev_b64 %" n ", 1, %" n "\n"
#define T_FMA32(n) "v_fma_f32 %" n ", %" n ", %8, %9\n"
#define T_MUL32(n) "v_mul_f32 %" n ", %" n ", %8\n"
#define T_RCP32(n) "v_rcp_f32 %" n ", %" n "\n"
#define T_CVT(n) "v_cvt_f64_i32 %" n ", %8\n"
#define T_READLANE(n) "v_readlane_b32 %" n ", %8, 3\n"
#define T_READFIRST(n) "v_readfirstlane_b32 %" n ", %8\n"
#define T_SMOV(n) "s_mov_b32 %" n ", 0x12345\n"
#define T_SADD(n) "s_add_u32 %" n ", %" n ", 7\n"
#define T_NOP(n) "s_nop 0\n"

KERNEL_HEAD(k_fma_f64) asm volatile(X64(T_FMA64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_mul_f64) asm volatile(X64(T_MUL64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_add_f64) asm volatile(X64(T_ADD64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_max_f64) asm volatile(X64(T_MAX64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_rcp_f64) asm volatile(X64(T_RCP64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_rsq_f64) asm volatile(X64(T_RSQ64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_cmp_f64) asm volatile(X64(T_CMP64) : S8(c) : "v"(a0), "v"(k)); KERNEL_TAIL
KERNEL_HEAD(k_cmp_i32) asm volatile(X64(T_CMP32) : S8(c) : "v"(i0), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_cndmask) asm volatile(X64(T_CNDMASK) : V8(i) : "v"(ik), "s"(c0)); KERNEL_TAIL
KERNEL_HEAD(k_mov_b32) asm volatile(X64(T_MOV32) : V8(i) : "v"(ik), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_and_b32) asm volatile(X64(T_AND32) : V8(i) : "v"(ik), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_add_u32) asm volatile(X64(T_ADD32) : V8(i) : "v"(ik), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_lshl_b64) asm volatile(X64(T_LSHL64) : V8(c) : "v"(ik), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_fma_f32) asm volatile(X64(T_FMA32) : V8(f) : "v"(fk), "v"(fk2)); KERNEL_TAIL
KERNEL_HEAD(k_mul_f32) asm volatile(X64(T_MUL32) : V8(f) : "v"(fk), "v"(fk2)); KERNEL_TAIL
KERNEL_HEAD(k_rcp_f32) asm volatile(X64(T_RCP32) : V8(f) : "v"(fk), "v"(fk2)); KERNEL_TAIL
KERNEL_HEAD(k_cvt_f64_i32) asm volatile(X64(T_CVT) : V8(a) : "v"(i0), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_readlane) asm volatile(X64(T_READLANE) : S8(s) : "v"(i0), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_readfirstlane) asm volatile(X64(T_READFIRST) : S8(s) : "v"(i0), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_s_mov) asm volatile(X64(T_SMOV) : S8(s) : "v"(i0), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_s_add) asm volatile(X64(T_SADD) : S8(s) : "v"(i0), "v"(ik) : "scc"); KERNEL_TAIL
KERNEL_HEAD(k_s_nop) asm volatile(X64(T_NOP) : S8(s) : "v"(i0), "v"(ik)); KERNEL_TAIL
// mixed 1:1 (32 + 32 per iteration): %0..%7 the VALU chains, %10..%17 the second class's registers
#define T_FMA64_SMOV(n) "v_fma_f64 %" n ", %" n ", %8, %9\ns_mov_b32 %1" n ", 0x12345\n"
#define T_AND32_SMOV(n) "v_and_b32 %" n ", %" n ", %8\ns_mov_b32 %1" n ", 0x12345\n"
#define T_FMA64_AND32(n) "v_fma_f64 %" n ", %" n ", %8, %9\nv_and_b32 %1" n ", %1" n ", %1" n "\n"
#define T_FMA64_READLANE(n) "v_fma_f64 %" n ", %" n ", %8, %9\nv_readlane_b32 %1" n ", %18, 3\n"
KERNEL_HEAD(k_fma_f64_x_s_mov) asm volatile(X32(T_FMA64_SMOV) : V8(a), "+v"(k), "+v"(k2), S8(s) : ); KERNEL_TAIL
KERNEL_HEAD(k_and_b32_x_s_mov) { int j0 = ik, j1 = ik; asm volatile(X32(T_AND32_SMOV) : V8(i), "+v"(j0), "+v"(j1), S8(s) : ); } KERNEL_TAIL
KERNEL_HEAD(k_fma_f64_x_and_b32) asm volatile(X32(T_FMA64_AND32) : V8(a), "+v"(k), "+v"(k2), V8(i) : ); KERNEL_TAIL
KERNEL_HEAD(k_fma_f64_x_readlane) { int j0 = ik; asm volatile(X32(T_FMA64_READLANE) : V8(a), "+v"(k), "+v"(k2), S8(s), "+v"(j0) : ); } KERNEL_TAIL
// DEPENDENT chains (round 6): all 64 instructions on ONE register chain (dep1), on two (dep2) or four (dep4) -- what a wavefront
// pays when the next instruction needs the previous result (the serial polynomial / Newton chains of atan2, sincos, rcp), i.e. the
// LATENCY of the class, against the issue cost above; and an LDS round trip (ds_read whose address is the previous read's result)
#define D1(S) S("0") S("0") S("0") S("0") S("0") S("0") S("0") S("0")
#define D2(S) S("0") S("1") S("0") S("1") S("0") S("1") S("0") S("1")
#define D4(S) S("0") S("1") S("2") S("3") S("0") S("1") S("2") S("3")
#define D1_64(S) D1(S) D1(S) D1(S) D1(S) D1(S) D1(S) D1(S) D1(S)
#define D2_64(S) D2(S) D2(S) D2(S) D2(S) D2(S) D2(S) D2(S) D2(S)
#define D4_64(S) D4(S) D4(S) D4(S) D4(S) D4(S) D4(S) D4(S) D4(S)
#define T_LDSCHASE(n) "ds_read_b32 %" n ", %" n "\ns_waitcnt lgkmcnt(0)\n"
KERNEL_HEAD(k_fma_f64_dep1) asm volatile(D1_64(T_FMA64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_fma_f64_dep2) asm volatile(D2_64(T_FMA64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_fma_f64_dep4) asm volatile(D4_64(T_FMA64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_add_f64_dep1) asm volatile(D1_64(T_ADD64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
KERNEL_HEAD(k_and_b32_dep1) asm volatile(D1_64(T_AND32) : V8(i) : "v"(ik), "v"(ik)); KERNEL_TAIL
KERNEL_HEAD(k_rcp_f64_dep1) asm volatile(D1_64(T_RCP64) : V8(a) : "v"(k), "v"(k2)); KERNEL_TAIL
__global__ void __launch_bounds__(64) k_lds_chase(Out *out, double seed) {
  __shared__ int lds[64];
  lds[threadIdx.x] = 0;  // every chain reads address 0, which holds 0
  __syncthreads();
  int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0, i6 = 0, i7 = 0;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < ITERS; ++it) asm volatile(D1_64(T_LDSCHASE) : V8(i) : : "memory");
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x].ticks = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
  if ((i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) == 12345 + (int)seed) out[blockIdx.x].sink = 1.0;
}
KERNEL_HEAD(k_empty_loop) asm volatile("" : V8(a)); KERNEL_TAIL

typedef void (*kern_t)(Out *, double);
struct Entry { const char *name; kern_t fn; int per_iter; };
#define E(n, c) {#n, k_##n, c}

int main(int argc, char **argv) {
  const Entry table[] = {E(fma_f64, 64), E(mul_f64, 64), E(add_f64, 64), E(max_f64, 64), E(cmp_f64, 64), E(cmp_i32, 64), E(cndmask, 64),
                         E(mov_b32, 64), E(and_b32, 64), E(add_u32, 64), E(lshl_b64, 64), E(fma_f32, 64), E(mul_f32, 64), E(rcp_f64, 64),
                         E(rsq_f64, 64), E(rcp_f32, 64), E(cvt_f64_i32, 64), E(readlane, 64), E(readfirstlane, 64), E(s_mov, 64), E(s_add, 64),
                         E(s_nop, 64), E(fma_f64_x_s_mov, 64), E(and_b32_x_s_mov, 64), E(fma_f64_x_and_b32, 64), E(fma_f64_x_readlane, 64), E(fma_f64_dep1, 64), E(fma_f64_dep2, 64), E(fma_f64_dep4, 64), E(add_f64_dep1, 64),
                         E(and_b32_dep1, 64), E(rcp_f64_dep1, 64), E(lds_chase, 64),
                         E(empty_loop, 0)};
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int simds = prop.multiProcessorCount * 4;
  Out *d_out;
  const int max_blocks = simds * 8;
  CHECK(hipMalloc(&d_out, sizeof(Out) * max_blocks));
  std::vector<Out> h(max_blocks);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_rate_khz\": %d, \"iters\": %d,\n \"note\": \"ticks = s_memtime (shader clock) per instruction per wave; ns_per_inst_per_simd = launch duration / (instructions per wave x waves per SIMD); cycles = ns x shader GHz with GHz = ticks / s_memrealtime (100 MHz)\",\n \"classes\": {\n",
         prop.name, prop.multiProcessorCount, simds, prop.clockRate, ITERS);
  const int n = sizeof(table) / sizeof(table[0]);
  // the loop's own cost (s_add / s_cmp / s_cbranch + anything the compiler adds), per iteration, per W
  double empty_ns[9] = {0}, empty_ticks[9] = {0};
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = 0; t < n; ++t) {
      const Entry &en = table[pass == 0 ? n - 1 : t];
      if (pass == 1 && t == n - 1) break;
      if (pass == 1) printf("  \"%s\": {", en.name);
      bool first = true;
      for (int W : {1, 2, 4, 8}) {
        const int blocks = simds * W;
        double best_ms = 1e30;
        std::vector<double> tick_med;
        double ghz = 0;
        for (int rep = 0; rep < 5; ++rep) {
          CHECK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(64), 0, 0, d_out, 1.0);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep == 0) continue;  // warm-up
          best_ms = std::min(best_ms, (double)ms);
          CHECK(hipMemcpy(h.data(), d_out, sizeof(Out) * blocks, hipMemcpyDeviceToHost));
          std::vector<double> tk(blocks);
          double sum_t = 0, sum_r = 0;
          for (int b = 0; b < blocks; ++b) { tk[b] = (double)h[b].ticks; sum_t += tk[b]; sum_r += (double)h[b].realtime; }
          std::nth_element(tk.begin(), tk.begin() + blocks / 2, tk.end());
          tick_med.push_back(tk[blocks / 2]);
          ghz = sum_t / sum_r * 0.1;  // s_memrealtime: 100 MHz
        }
        std::sort(tick_med.begin(), tick_med.end());
        const double ticks_iter = tick_med[tick_med.size() / 2] / ITERS, ns_iter = best_ms * 1e6 / ((double)ITERS * W);
        if (pass == 0) { empty_ns[W] = ns_iter; empty_ticks[W] = ticks_iter; continue; }
        const double ticks_inst = (ticks_iter - empty_ticks[W]) / en.per_iter, ns_inst = (ns_iter - empty_ns[W]) / en.per_iter;
        printf("%s\"w%d\": {\"ticks_per_inst_per_wave\": %.3f, \"ns_per_inst_per_simd\": %.4f, \"cycles_per_inst_per_simd\": %.3f, \"shader_ghz\": %.3f}",
               first ? "" : ", ", W, ticks_inst, ns_inst, ns_inst * ghz, ghz);
        first = false;
      }
      if (pass == 0) break;
      printf("}%s\n", t == n - 2 ? "" : ",");
    }
  }
  printf(" },\n \"empty_loop_ns_per_iter_per_simd\": {\"w1\": %.3f, \"w2\": %.3f, \"w4\": %.3f, \"w8\": %.3f},\n \"empty_loop_ticks_per_iter\": {\"w1\": %.2f, \"w2\": %.2f, \"w4\": %.2f, \"w8\": %.2f}\n}\n",
         empty_ns[1], empty_ns[2], empty_ns[4], empty_ns[8], empty_ticks[1], empty_ticks[2], empty_ticks[4], empty_ticks[8]);
  return 0;
}
