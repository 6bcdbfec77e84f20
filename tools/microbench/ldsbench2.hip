#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(1))) const float glb_float_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
// wave 4 times 16 LDS-direct loads while waves 0..3 run background work: BG bit0 = MFMA, bit1 = ds_read
template <int BG>
__global__ __launch_bounds__(320) void k(const float* g, long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    f32x16 acc = {0};
    float a = lane, b = 1.f;
    for (int it = 0; it < iters; ++it) {
      if (BG & 2) { a += s[8192 + ((it * 64 + lane) & 4095)]; b += s[8192 + ((it * 64 + lane + 2048) & 4095)]; }
      if (BG & 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
      }
    }
    if (acc[0] + a + b == 123.456f) sink[0] = acc[1];
    return;
  }
  const float* src = g + (size_t)blockIdx.x * 65536 + lane * 4;
  // let the background get going
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 16; ++i) __builtin_amdgcn_global_load_lds((glb_float_t*)(src + i * 256), (lds_float_t*)(s + i * 256), 16, 0, 0);
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
}
int main() {
  float* g; long long* out; float* sink;
  hipMalloc(&g, 1024 * 65536 * 4); hipMemset(g, 0, 1024 * 65536 * 4); hipMalloc(&out, 64); hipMalloc(&sink, 64);
  long long h[2];
  const char* names[] = {"idle", "MFMA", "ds_read", "MFMA+ds_read"};
  for (int bg = 0; bg < 4; ++bg) for (int blocks : {1, 512}) for (int rep = 0; rep < 2; ++rep) {
    const int iters = 200000;
    if (bg == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (bg == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (bg == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (bg == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    hipDeviceSynchronize(); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    if (rep) printf("background %-14s blocks %4d: issue 16 x4 loads %6lld cycles, then wait %6lld\n", names[bg], blocks, h[0], h[1]);
  }
  return 0;
}
