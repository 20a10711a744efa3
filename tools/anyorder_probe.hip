// anyorder_probe.hip -- does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let the next kernel of a stream start while
// the previous one drains (gfx950, ROCm 7.2)?  Two kernels of busy-waiting workgroups stamp wall_clock64() at entry and
// exit; the host prints how far kernel B's first entry lies before kernel A's last exit, with and without the flag.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_busy(long long *t_in, long long *t_out, int ticks, int lds_pad) {
  extern __shared__ char pad[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) t_in[blockIdx.x] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (lds_pad < 0) pad[threadIdx.x] = 1;
  if (threadIdx.x == 0) t_out[blockIdx.x] = wall_clock64();
}

int main() {
  const int GA = 256 * 8 * 2 + 700, GB = 256 * 8;          // A: 2.x generations at 8 WGs/CU
  long long *d; hipMalloc(&d, sizeof(long long) * 2 * (GA + GB));
  long long *ain = d, *aout = d + GA, *bin = d + 2 * GA, *bout = bin + GB;
  hipStream_t st; hipStreamCreate(&st);
  int ticks = 1000;                                          // 100 MHz clock: 10 us per workgroup
  for (int mode = 0; mode < 3; mode++) {
    const int flags = mode == 1 ? hipExtAnyOrderLaunch : 0;
    for (int rep = 0; rep < 3; rep++) {
      if (mode == 2) {                                       // reference: plain <<<>>> launches
        k_busy<<<GA, 256, 0, st>>>(ain, aout, ticks, 0);
        k_busy<<<GB, 256, 0, st>>>(bin, bout, ticks, 0);
      } else {
        hipExtLaunchKernelGGL(k_busy, dim3(GA), dim3(256), 0, st, nullptr, nullptr, flags, ain, aout, ticks, 0);
        hipExtLaunchKernelGGL(k_busy, dim3(GB), dim3(256), 0, st, nullptr, nullptr, flags, bin, bout, ticks, 0);
      }
      hipStreamSynchronize(st);
    }
    std::vector<long long> h(2 * (GA + GB));
    hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    const long long a0 = *std::min_element(h.begin(), h.begin() + GA), a1 = *std::max_element(h.begin() + GA, h.begin() + 2 * GA);
    const long long b0 = *std::min_element(h.begin() + 2 * GA, h.begin() + 2 * GA + GB), b1 = *std::max_element(h.begin() + 2 * GA + GB, h.end());
    std::vector<long long> as(h.begin(), h.begin() + GA); std::sort(as.begin(), as.end());
    printf("mode %d (%s): A %.1f us, B starts %.2f us %s A's last exit, total %.1f us; A's last generation began %.1f us before A ended\n", mode,
           mode == 0 ? "hipExt, no flag" : mode == 1 ? "hipExtAnyOrderLaunch" : "<<<>>>", (a1 - a0) / 100.0,
           (b0 > a1 ? b0 - a1 : a1 - b0) / 100.0, b0 > a1 ? "AFTER" : "BEFORE", (b1 - a0) / 100.0, (a1 - as[GA - 1]) / 100.0);
  }
  return 0;
}
