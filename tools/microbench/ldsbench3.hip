#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(1))) const float glb_float_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
// wave 4 times 16 operations while waves 0..3 run MFMA + ds_read (two workgroups per CU at 512 blocks)
// MODE 0: 16 LDS-direct x4 loads; 1: same + 3 VALU each; 2: 48 VALU only; 3: 96 SALU only; 4: loads with saddr form
template <int MODE>
__global__ __launch_bounds__(320) void k(const float* g, long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    f32x16 acc = {0};
    float a = lane, b = 1.f;
    for (int it = 0; it < iters; ++it) {
      a += s[8192 + ((it * 64 + lane) & 4095)]; b += s[8192 + ((it * 64 + lane + 2048) & 4095)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    }
    if (acc[0] + a + b == 123.456f) sink[0] = acc[1];
    return;
  }
  const float* base = g + (size_t)blockIdx.x * 65536;
  const float* src = base + lane * 4;
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
  int v = lane; int sc = blockIdx.x;
  long long t0 = __builtin_readcyclecounter();
  if (MODE == 0 || MODE == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_global_load_lds((glb_float_t*)(src + i * 256), (lds_float_t*)(s + i * 256), 16, 0, 0);
      if (MODE == 1) { asm volatile("v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 5" : "+v"(v)); }
    }
  } else if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3\n v_add_u32 %0, %0, 5" : "+v"(v));
  } else if (MODE == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 3\n s_add_u32 %0, %0, 5" : "+s"(sc));
  } else if (MODE == 4) {
    const unsigned voff = lane * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_global_load_lds((glb_float_t*)((const char*)(base + i * 256) + voff), (lds_float_t*)(s + i * 256), 16, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
  if (v + sc == 123456789) sink[1] = 1.f;
}
int main() {
  float* g; long long* out; float* sink;
  hipMalloc(&g, 1024 * 65536 * 4); hipMemset(g, 0, 1024 * 65536 * 4); hipMalloc(&out, 64); hipMalloc(&sink, 64);
  long long h[2];
  const char* names[] = {"16 x4 loads", "16 x4 loads + 3 VALU each", "48 VALU", "96 SALU", "16 x4 loads saddr form"};
  for (int m = 0; m < 5; ++m) for (int rep = 0; rep < 2; ++rep) {
    const int iters = 200000, blocks = 512;
    if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(320), 65536, 0, g, out, sink, iters);
    hipDeviceSynchronize(); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    if (rep) printf("%-28s: %6lld cycles, then wait %6lld\n", names[m], h[0], h[1]);
  }
  return 0;
}
