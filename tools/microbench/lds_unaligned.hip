// Does global_load_lds_dwordx4 accept global addresses that are only 4-byte aligned?  (prints per-shift correctness)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(1))) const float glb_float_t;
__global__ void k(const float* g, float* out, int shift) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((glb_float_t*)(g + shift + lane * 4), (lds_float_t*)s, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = s[lane * 4 + i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096); hipMalloc(&o, 1024); hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 5; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, g, o, shift);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
    printf("shift %d floats: %s, %d mismatches (first values %g %g %g %g)\n", shift, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
