// How long does the dispatcher need to launch N workgroups of trivial work?  (wave-launch overhead)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSB>
__global__ __launch_bounds__(256) void k_empty(float *out, int n) {
  __shared__ float lds[LDSB / 4];
  if (threadIdx.x == 0) lds[0] = 1.0f;
  if (n < 0) out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x % (LDSB / 4)];
}
__global__ __launch_bounds__(64) void k_empty64(float *out, int n) {
  if (n < 0) out[blockIdx.x * 64 + threadIdx.x] = 1.0f;
}
template <typename F> float time_us(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 20; i++) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / 20;
}
int main() {
  float *out; (void)hipMalloc(&out, 1 << 26);
  for (int wgs : {256, 1024, 4096, 16384, 65536}) {
    printf("256-thread WGs %6d: LDS 1KB %7.2f us   LDS 22KB %7.2f us   LDS 44KB %7.2f us | 64-thread WGs x4: %7.2f us\n", wgs,
           time_us([&] { k_empty<1024><<<wgs, 256>>>(out, 1); }), time_us([&] { k_empty<22528><<<wgs, 256>>>(out, 1); }),
           time_us([&] { k_empty<45056><<<wgs, 256>>>(out, 1); }), time_us([&] { k_empty64<<<wgs * 4, 64>>>(out, 1); }));
  }
  return 0;
}
