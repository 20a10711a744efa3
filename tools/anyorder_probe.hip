// tools/anyorder_probe.hip -- does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on this device?  Two one-workgroup
// kernels of ~50 us each, the second launched with / without the flag; prints the pair's time.   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int *out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) { }
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  int *d; hipMalloc(&d, 1024);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long cyc = 100000;                              // ~50 us at ~2 GHz
  for (int flag = 0; flag < 2; flag++) {
    for (int rep = 0; rep < 3; rep++) {
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cyc, d);
      if (flag) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 1);
      else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cyc, d + 1);
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::printf("second kernel %s: pair took %.1f us (err %d)\n", flag ? "hipExtAnyOrderLaunch" : "in order", ms * 1e3, (int)hipGetLastError());
    }
  }
  return 0;
}
