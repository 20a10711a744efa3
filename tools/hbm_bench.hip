// hbm_bench.hip -- what does this GPU sustain for k_demod's traffic shape?  Streams R bytes in (16-byte loads, a wave
// per 5 KB "burst") and writes W bytes out per burst, nothing else; working sets beyond the 256 MB memory-side cache.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_bench.hip -o tools/hbm_bench.bin && tools/hbm_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NLD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_stream(const float4 *__restrict__ in, float *__restrict__ out, int B, int stride4, int nout, int ostride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * WAVES + wave;
  if (b >= B) return;
  const float4 *p = in + (size_t)b * stride4;
  float4 v[NLD];
#pragma unroll
#ifdef NT_LOAD
  for (int i = 0; i < NLD; i++) { typedef float f4v __attribute__((ext_vector_type(4))); f4v t = {0, 0, 0, 0}; if (lane + 64 * i < stride4) t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p) + lane + 64 * i); v[i] = make_float4(t.x, t.y, t.z, t.w); }
#else
  for (int i = 0; i < NLD; i++) v[i] = (lane + 64 * i < stride4) ? p[lane + 64 * i] : make_float4(0, 0, 0, 0);
#endif
  float s = 0;
#pragma unroll
  for (int i = 0; i < NLD; i++) s += v[i].x + v[i].y + v[i].z + v[i].w;
#ifdef EXTRA_VALU                                            // how much arithmetic rides for free under this traffic?
  {
    float a0 = s, a1 = s + 1.0f, a2 = s + 2.0f, a3 = s + 3.0f;
#pragma unroll 16
    for (int k = 0; k < EXTRA_VALU / 4; k++) {
      asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(1.0001f));
    }
    s = a0 + a1 + a2 + a3;
  }
#endif
  float *o = out + (size_t)b * ostride;
#ifdef NT_STORE
  for (int m = lane; m < nout; m += 64) __builtin_nontemporal_store(s + m, o + m);
#else
  for (int m = lane; m < nout; m += 64) o[m] = s + m;
#endif
}

int main() {
  const int B = 65536 * 3, stride4 = 5000 / 16;            // 3 batches of 64 K bursts (1 GB): no cross-launch cache reuse
  float4 *in; float *out;
  (void)hipMalloc(&in, (size_t)B * stride4 * 16); (void)hipMalloc(&out, (size_t)B * 160 * 4);
  (void)hipMemset(in, 0, (size_t)B * stride4 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int cases[][2] = {{0, 160}, {148, 148}, {148, 160}, {160, 160}, {128, 128}, {32, 32}};
  for (auto &cs : cases) {
    const int nout = cs[0], ostride = cs[1];
    for (int waves : {4}) {
      auto launch = [&](int k) {
        const float4 *p = in + (size_t)(k % 3) * 65536 * stride4;
        float *o = out + (size_t)(k % 3) * 65536 * 160;
        if (waves == 4) k_stream<5, 4><<<65536 / 4, 256>>>(p, o, 65536, stride4, nout, ostride);
        else k_stream<5, 8><<<65536 / 8, 512>>>(p, o, 65536, stride4, nout, ostride);
      };
      for (int k = 0; k < 300; k++) launch(k);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      const int K = 600;
      for (int k = 0; k < K; k++) launch(k);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1000 / K, bytes = 65536.0 * (stride4 * 16 + nout * 4);
      std::printf("read %d B + write %d B (stride %d B) per burst, %d waves/WG: %.1f us per 64 K bursts = %.2f TB/s\n", stride4 * 16, nout * 4, ostride * 4, waves, us, bytes / us / 1e6);
    }
  }
  return 0;
}
