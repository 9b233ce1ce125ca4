// valu_rate.hip -- how many cycles does a SIMD of gfx950 need per wave64 vector instruction of the kinds the walk is
// made of (integer add / and / compare + select / 24-bit multiply-add), as a function of the waves resident per SIMD?
// MI355X_MICROARCH.md quotes 2 cycles for v_fma_f32 once more than one wave is resident.
// Every wave runs REPS x 64 instructions on 8 independent registers and stamps s_memtime around them.
//   build: hipcc --offload-arch=gfx950 -O3 -o ab/valu_rate scripts/micro/valu_rate.hip      run: ab/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REPS 2000

#define OP8(ins)                                                                      \
  asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n"  \
               ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n"  \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(unsigned long long *out, unsigned *sink, unsigned k) {
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (KIND == 0) OP8("v_add_u32")
      if (KIND == 1) OP8("v_and_b32")
      if (KIND == 2) OP8("v_mul_u32_u24")
      if (KIND == 3) {  // compare + select pairs (the walk's predicates)
        asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                     "v_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
      }
      if (KIND == 4) {  // float fma for reference (the guide's 2-cycle instruction)
        asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
      }
      if (KIND == 5) {  // one dependent chain: latency of back-to-back dependent adds
        asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n" : "+v"(a0) : "v"(k));
      }
      if (KIND == 6) {  // vector compare -> scalar and -> select: the VALU -> SALU -> VALU hand-over (6 instructions)
        asm volatile("v_cmp_lt_u32 vcc, %0, %2\ns_and_b64 vcc, vcc, exec\nv_cndmask_b32 %0, %0, %2, vcc\n"
                     "v_cmp_lt_u32 vcc, %1, %2\ns_and_b64 vcc, vcc, exec\nv_cndmask_b32 %1, %1, %2, vcc\n"
                     : "+v"(a0), "+v"(a1) : "v"(k) : "vcc", "scc");
      }
      if (KIND == 7) {  // the same without the scalar instruction (4 instructions)
        asm volatile("v_cmp_lt_u32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %2, vcc\n"
                     "v_cmp_lt_u32 vcc, %1, %2\nv_cndmask_b32 %1, %1, %2, vcc\n"
                     : "+v"(a0), "+v"(a1) : "v"(k) : "vcc");
      }
      if (KIND == 8) {  // scalar chain: 8 dependent s_and_b64 / s_or_b64
        asm volatile("s_and_b64 vcc, vcc, exec\ns_or_b64 vcc, vcc, exec\ns_and_b64 vcc, vcc, exec\ns_or_b64 vcc, vcc, exec\n"
                     "s_and_b64 vcc, vcc, exec\ns_or_b64 vcc, vcc, exec\ns_and_b64 vcc, vcc, exec\ns_or_b64 vcc, vcc, exec\n" ::: "vcc", "scc");
      }
      if (KIND == 9) {  // compare into a scalar pair, two scalar ops on it, select (the walk's predicate pattern; 8 instructions)
        asm volatile("v_cmp_lt_u32 s[20:21], %0, %2\nv_cmp_gt_u32 s[22:23], %1, %2\ns_and_b64 s[20:21], s[20:21], s[22:23]\n"
                     "v_cndmask_b32 %0, %0, %2, s[20:21]\n"
                     "v_cmp_lt_u32 s[24:25], %1, %2\nv_cmp_gt_u32 s[26:27], %0, %2\ns_or_b64 s[24:25], s[24:25], s[26:27]\n"
                     "v_cndmask_b32 %1, %1, %2, s[24:25]\n"
                     : "+v"(a0), "+v"(a1) : "v"(k) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
      }
      if (KIND == 10) {  // a taken uniform branch per 4 vector instructions
        asm volatile("v_add_u32 %0, %0, %2\nv_add_u32 %1, %1, %2\ns_cbranch_scc1 1f\n1:\nv_add_u32 %0, %0, %2\nv_add_u32 %1, %1, %2\n"
                     "s_cmp_eq_u32 s20, s20\ns_cbranch_scc1 2f\ns_nop 0\n2:\n"
                     : "+v"(a0), "+v"(a1) : "v"(k) : "s20", "scc");
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) sink[0] = a0;
}

template <int KIND>
static void run(const char *name, int per_rep) {
  unsigned long long *d;
  unsigned *sink;
  hipMalloc(&d, 8 * 8192 * 8);
  hipMalloc(&sink, 64);
  printf("%-34s", name); fflush(stdout);
  for (int w : {1, 2, 3, 4, 6, 8}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d, sink, 3u);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d, sink, 3u);
    if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed: %s", hipGetErrorString(hipGetLastError())); break; }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[blocks / 2];
    // cycles per instruction of one wave, and per instruction of the SIMD (w waves share it)
    printf("  w=%d: %5.2f /wave %5.2f /simd", w, cyc / ((double)REPS * per_rep), cyc / ((double)REPS * per_rep * w));
  }
  printf("\n");
  hipFree(d); hipFree(sink);
}

int main() {
  printf("cycles (s_memtime) per wave64 instruction: per wave, and per SIMD with w waves resident per SIMD\n");
  run<0>("v_add_u32 (8 independent)", 64);
  run<1>("v_and_b32", 64);
  run<2>("v_mul_u32_u24", 64);
  run<3>("v_cmp + v_cndmask (vcc)", 64);
  run<4>("v_fma_f32", 64);
  run<5>("v_add_u32 dependent chain", 64);
  run<6>("v_cmp -> s_and -> v_cndmask (vcc)", 48);
  run<7>("v_cmp -> v_cndmask (vcc), dependent", 32);
  run<8>("s_and/s_or dependent chain", 64);
  run<9>("2 v_cmp_e64, s_op, v_cndmask", 64);
  run<10>("4 v_add + 2 taken branches + s_cmp", 56);
  return 0;
}
