// VALU issue-rate microbenchmark for gfx950: how many cycles does a wave64 v_add_f32 / v_mul_f32 /
// v_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 hold a SIMD for, at 1..8 waves per SIMD?
// hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (OP == 0) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == 1) {
        asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == 2) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void kp(float *out, int iters) {
  float2v a0 = {(float)threadIdx.x, 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float2v b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      if (OP == 0) {
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
      } else if (OP == 1) {
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
      } else {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
      }
    }
  }
  float2v s = a0 + a1 + a2 + a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename F>
double time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate / 1e6;
  printf("%s CUs %d clock %.2f GHz\n", p.gcnArchName, cus, ghz);
  float *out; hipMalloc(&out, sizeof(float) * cus * 2048 * 4);
  const int iters = 20000;
  const char *names[6] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"};
  for (int wps = 1; wps <= 8; wps *= 2) {         // waves per SIMD
    const int threads = 256, blocks = cus * wps;  // 4 waves per block = 1 per SIMD
    double ms[6];
    ms[0] = time_ms([&] { k<0><<<blocks, threads>>>(out, iters); });
    ms[1] = time_ms([&] { k<1><<<blocks, threads>>>(out, iters); });
    ms[2] = time_ms([&] { k<2><<<blocks, threads>>>(out, iters); });
    ms[3] = time_ms([&] { kp<0><<<blocks, threads>>>(out, iters); });
    ms[4] = time_ms([&] { kp<1><<<blocks, threads>>>(out, iters); });
    ms[5] = time_ms([&] { kp<2><<<blocks, threads>>>(out, iters); });
    for (int i = 0; i < 6; i++) {
      const double insts_per_simd = (double)iters * REP * wps;      // wave-instructions issued per SIMD
      const double ns_per_inst = ms[i] * 1e6 / insts_per_simd;
      printf("waves/SIMD %d  %-13s %8.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles at %.2f GHz)\n", wps,
             names[i], ms[i], ns_per_inst, ns_per_inst * ghz, ghz);
    }
  }
  return 0;
}
