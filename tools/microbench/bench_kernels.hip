// Kernel micro-benchmark entry point (tuning aid, not part of the product path): times one
// conv_tc variant — including ablated builds — on synthetic resident data with HIP events.
#include <cstring>
#include <vector>

#include "conv_tc_dma_kernel.h"
#include "conv_tc_kernel.h"
#include "conv_tc_pp_kernel.h"
#include "resblock_pair_kernel.h"

namespace evmi {

struct Variant {
  const char* name;
  int c_in, ks, max_dil;
  ConvTcLaunch launch;
};

//        name, CIN KC BM BN WM WN KS TAPS MAXDIL ABL
#define EVMI_VARIANTS(X)                                           \
  X("c128k11_base", 128, 64, 128, 128, 2, 2, 11, 1, 5, 0)          \
  X("c128k11_noA", 128, 64, 128, 128, 2, 2, 11, 1, 5, 1)           \
  X("c128k11_noX", 128, 64, 128, 128, 2, 2, 11, 1, 5, 2)           \
  X("c128k11_noEpi", 128, 64, 128, 128, 2, 2, 11, 1, 5, 4)         \
  X("c128k11_noMfma", 128, 64, 128, 128, 2, 2, 11, 1, 5, 8)        \
  X("c128k11_noA_noX", 128, 64, 128, 128, 2, 2, 11, 1, 5, 3)       \
  X("c128k11_noA_noX_noEpi", 128, 64, 128, 128, 2, 2, 11, 1, 5, 7) \
  X("c128k11_bm64", 128, 64, 64, 128, 1, 4, 11, 1, 5, 0)           \
  X("c128k11_bn256", 128, 64, 128, 256, 2, 4, 11, 1, 5, 0)         \
  X("c128k11_kc128", 128, 128, 128, 128, 2, 2, 11, 1, 5, 0)        \
  X("c128k11_kc128_bn256", 128, 128, 128, 256, 2, 4, 11, 1, 5, 0)  \
  X("c128k3_base", 128, 64, 128, 128, 2, 2, 3, 1, 5, 0)            \
  X("c128k3_noX", 128, 64, 128, 128, 2, 2, 3, 1, 5, 2)             \
  X("c128k3_noEpi", 128, 64, 128, 128, 2, 2, 3, 1, 5, 4)           \
  X("c128k3_kc128", 128, 128, 128, 128, 2, 2, 3, 1, 5, 0)          \
  X("c128k3_kc128_t3", 128, 128, 128, 128, 2, 2, 3, 3, 5, 0)       \
  X("c128k3_bm64_kc128_t3", 128, 128, 64, 128, 1, 4, 3, 3, 5, 0)   \
  X("c64k11_t1", 64, 64, 64, 128, 1, 4, 11, 1, 5, 0)               \
  X("c64k3_base", 64, 64, 64, 128, 1, 4, 3, 3, 5, 0)               \
  X("c64k3_bn256", 64, 64, 64, 256, 1, 4, 3, 3, 5, 0)              \
  X("c32k11_base", 32, 32, 32, 256, 1, 4, 11, 11, 5, 0)            \
  X("c32k11_noEpi", 32, 32, 32, 256, 1, 4, 11, 11, 5, 4)           \
  X("c32k11_bn512", 32, 32, 32, 512, 1, 4, 11, 11, 5, 0)           \
  X("c32k11_bn512_w8", 32, 32, 32, 512, 1, 8, 11, 11, 5, 0)        \
  X("c32k3_base", 32, 32, 32, 256, 1, 4, 3, 3, 5, 0)               \
  X("c32k3_bn512_w8", 32, 32, 32, 512, 1, 8, 3, 3, 5, 0)           \
  X("c256k11_base", 256, 64, 128, 128, 2, 2, 11, 1, 5, 0)          \
  X("c256k11_kc128", 256, 128, 128, 128, 2, 2, 11, 1, 5, 0)        \
  X("c256k11_bn256", 256, 64, 128, 256, 2, 4, 11, 1, 5, 0)         \
  X("c256k11_bm256", 256, 64, 256, 128, 4, 2, 11, 1, 5, 0)         \
  X("c128k11_bn256_wm1", 128, 64, 128, 256, 1, 8, 11, 1, 5, 0)     \
  X("c128k11_bn256_wm4", 128, 64, 128, 256, 4, 2, 11, 1, 5, 0)     \
  X("c128k11_bn512", 128, 64, 128, 512, 2, 8, 11, 1, 5, 0)         \
  X("c128k11_bn256_noEpi", 128, 64, 128, 256, 2, 4, 11, 1, 5, 4)   \
  X("c128k11_bn256_noX", 128, 64, 128, 256, 2, 4, 11, 1, 5, 2)     \
  X("c128k11_bn256_noA", 128, 64, 128, 256, 2, 4, 11, 1, 5, 1)     \
  X("c128k11_bn256_noMfma", 128, 64, 128, 256, 2, 4, 11, 1, 5, 8)  \
  X("c128k11_bn256_prio", 128, 64, 128, 256, 2, 4, 11, 1, 5, 32)   \
  X("c256k11_bn256_prio", 256, 64, 128, 256, 2, 4, 11, 1, 5, 32)   \
  X("c128k3_bn256_prio", 128, 64, 128, 256, 2, 4, 3, 1, 5, 32)     \
  X("c128k7_bn256", 128, 64, 128, 256, 2, 4, 7, 1, 5, 0)           \
  X("c128k3_bn256", 128, 64, 128, 256, 2, 4, 3, 1, 5, 0)           \
  X("c128k3_bn512", 128, 64, 128, 512, 2, 8, 3, 1, 5, 0)           \
  X("c64k11_bn256_w8_t1", 64, 64, 64, 256, 1, 8, 11, 1, 5, 0)      \
  X("c64k11_bn512_w8_t1", 64, 64, 64, 512, 1, 8, 11, 1, 5, 0)      \
  X("c64k11_bn512_w16_t1", 64, 64, 64, 512, 1, 16, 11, 1, 5, 0)    \
  X("c64k3_bn256_w8_t3", 64, 64, 64, 256, 1, 8, 3, 3, 5, 0)        \
  X("c64k3_bn512_w16_t3", 64, 64, 64, 512, 1, 16, 3, 3, 5, 0)      \
  X("c64k7_bn256_w8_t1", 64, 64, 64, 256, 1, 8, 7, 1, 5, 0)        \
  X("c32k11_bn1024_w16", 32, 32, 32, 1024, 1, 16, 11, 11, 5, 0)    \
  X("c32k3_bn1024_w16", 32, 32, 32, 1024, 1, 16, 3, 3, 5, 0)       \
  X("c256k11_bn256_bm256", 256, 64, 256, 256, 4, 4, 11, 1, 5, 0)   \
  X("c256k3_bn256", 256, 64, 128, 256, 2, 4, 3, 1, 5, 0)           \
  X("c256k7_bn256", 256, 64, 128, 256, 2, 4, 7, 1, 5, 0)           \
  X("c128k11_bn512_wm1", 128, 64, 128, 512, 1, 8, 11, 1, 5, 0)     \
  X("c128k7_bn512_wm1", 128, 64, 128, 512, 1, 8, 7, 1, 5, 0)       \
  X("c128k3_bn512_wm1", 128, 64, 128, 512, 1, 8, 3, 1, 5, 0)       \
  X("c256k11_bn512_wm1", 256, 64, 128, 512, 1, 8, 11, 1, 5, 0)     \
  X("c256k11_bm256_mt4", 256, 64, 256, 256, 2, 4, 11, 1, 5, 0)     \
  X("c256k3_bn512_wm1", 256, 64, 128, 512, 1, 8, 3, 1, 5, 0)       \
  X("c128k11_bn256_wm1_w4", 128, 64, 128, 256, 1, 4, 11, 1, 5, 0)  \
  X("c128k11_bn512_nt4", 128, 64, 128, 512, 2, 4, 11, 1, 5, 0)     \
  X("c128k7_bn512_nt4", 128, 64, 128, 512, 2, 4, 7, 1, 5, 0)       \
  X("c128k3_bn512_nt4", 128, 64, 128, 512, 2, 4, 3, 1, 5, 0)       \
  X("c256k11_bn512_nt4", 256, 64, 128, 512, 2, 4, 11, 1, 5, 0)     \
  X("c256k3_bn512_nt4", 256, 64, 128, 512, 2, 4, 3, 1, 5, 0)

#define EVMI_VARIANTS_OCC(X)                                              \
  X("c128k11_bn256_occ4", 128, 64, 128, 256, 2, 4, 11, 1, 5, 0, 4)      \
  X("c128k11_bn128_occ3", 128, 64, 128, 128, 2, 2, 11, 1, 5, 0, 3)      \
  X("c256k11_bn256_occ4", 256, 64, 128, 256, 2, 4, 11, 1, 5, 0, 4)      \
  X("c128k11_bn256_md1_occ4", 128, 64, 128, 256, 2, 4, 11, 1, 1, 0, 4)  \
  X("c128k11_bn256_md1_occ2", 128, 64, 128, 256, 2, 4, 11, 1, 1, 0, 2)  \
  X("c128k11_bn256_md1_occ3", 128, 64, 128, 256, 2, 4, 11, 1, 1, 0, 3)  \
  X("c128k3_bn256_md1_occ4", 128, 64, 128, 256, 2, 4, 3, 1, 1, 0, 4)    \
  X("c128k3_bn256_md1_occ2", 128, 64, 128, 256, 2, 4, 3, 1, 1, 0, 2)    \
  /* round 2: two independent 256-thread blocks per CU (4 waves each, 64 x 128 or 128 x 64 per wave) */ \
  X("c128k11_w4_mt2nt4_occ2", 128, 64, 128, 256, 2, 2, 11, 1, 5, 0, 2)  \
  X("c128k11_w4_mt4nt2_occ2", 128, 64, 128, 256, 1, 4, 11, 1, 5, 0, 2)  \
  X("c128k11_w4_mt2nt4_occ1", 128, 64, 128, 256, 2, 2, 11, 1, 5, 0, 1)  \
  X("c128k7_w4_mt2nt4_occ2", 128, 64, 128, 256, 2, 2, 7, 1, 5, 0, 2)    \
  X("c128k3_w4_mt2nt4_occ2", 128, 64, 128, 256, 2, 2, 3, 1, 5, 0, 2)    \
  X("c256k11_w4_mt2nt4_occ2", 256, 64, 128, 256, 2, 2, 11, 1, 5, 0, 2)  \
  X("c256k3_w4_mt2nt4_occ2", 256, 64, 128, 256, 2, 2, 3, 1, 5, 0, 2)    \
  X("c128k11_w4_bn128_occ2", 128, 64, 128, 128, 2, 2, 11, 1, 5, 0, 2)   \
  X("c128k11_w4_bn128_occ3", 128, 64, 128, 128, 2, 2, 11, 1, 5, 0, 3)   \
  X("c128k11_bn256_tl", 128, 64, 128, 256, 2, 4, 11, 1, 5, 128, 2)      \
  X("c128k11_w4_mt2nt4_occ2_tl", 128, 64, 128, 256, 2, 2, 11, 1, 5, 128, 2)

static const std::vector<Variant>& variants() {
  static const std::vector<Variant> v = {
#define X(name, cin, kc, bm, bn, wm, wn, ks, taps, md, abl, occ) \
  Variant{name, cin, ks, md, make_conv_tc_launch<ConvTcCfg<cin, kc, bm, bn, wm, wn, ks, taps, md, abl, occ>>(name)},
      EVMI_VARIANTS_OCC(X)
#undef X
#define X(name, cin, kc, bm, bn, wm, wn, ks, taps, md, abl) \
  Variant{name, cin, ks, md, make_conv_tc_launch<ConvTcCfg<cin, kc, bm, bn, wm, wn, ks, taps, md, abl>>(name)},
      EVMI_VARIANTS(X)
#undef X
#define X(name, cin, ks, md, dbg, var) Variant{name, cin, ks, md, make_conv_dma_launch<ConvDmaCfg<cin, ks, md, dbg, var>>(name)},
      Variant{"c128k11_dma_w16", 128, 11, 5, make_conv_dma_launch<ConvDmaCfg<128, 11, 5, 0, 0, 8>>("c128k11_dma_w16")},
      Variant{"c256k11_dma_w16", 256, 11, 5, make_conv_dma_launch<ConvDmaCfg<256, 11, 5, 0, 0, 8>>("c256k11_dma_w16")},
      Variant{"c128k3_dma_w16", 128, 3, 5, make_conv_dma_launch<ConvDmaCfg<128, 3, 5, 0, 0, 8>>("c128k3_dma_w16")},
      X("c128k11_dma", 128, 11, 5, 0, 0) X("c128k7_dma", 128, 7, 5, 0, 0) X("c128k3_dma", 128, 3, 5, 0, 0)
      X("c256k11_dma", 256, 11, 5, 0, 0) X("c256k7_dma", 256, 7, 5, 0, 0) X("c256k3_dma", 256, 3, 5, 0, 0)
      X("c128k11_dma_tl", 128, 11, 5, 1, 0)
      X("c128k11_dma_v1", 128, 11, 5, 0, 1) X("c128k11_dma_v2", 128, 11, 5, 0, 2) X("c128k11_dma_v3", 128, 11, 5, 0, 3)
      X("c128k11_dma_v4", 128, 11, 5, 0, 4) X("c128k11_dma_v6", 128, 11, 5, 0, 6) X("c128k11_dma_v7", 128, 11, 5, 0, 7)
      X("c128k11_dma_a16", 128, 11, 5, 0, 16) X("c128k11_dma_a48", 128, 11, 5, 0, 48)
      X("c128k11_dma_a64", 128, 11, 5, 0, 64) X("c128k11_dma_a128", 128, 11, 5, 0, 128) X("c128k11_dma_a192", 128, 11, 5, 0, 192)
      X("c128k11_dma_v1024", 128, 11, 5, 0, 1024) X("c128k11_dma_v1025", 128, 11, 5, 0, 1025) X("c256k11_dma_v1024", 256, 11, 5, 0, 1024)
      X("c128k7_dma_v1024", 128, 7, 5, 0, 1024)
      X("c128k11_dma_v2048", 128, 11, 5, 0, 2048) X("c256k11_dma_v2048", 256, 11, 5, 0, 2048)
      X("c128k11_dma_a208", 128, 11, 5, 0, 208) X("c128k11_dma_a256", 128, 11, 5, 0, 256) X("c128k11_dma_a512", 128, 11, 5, 0, 512)
      X("c256k11_dma_v1", 256, 11, 5, 0, 1) X("c256k11_dma_v2", 256, 11, 5, 0, 2) X("c256k11_dma_v3", 256, 11, 5, 0, 3)
#undef X
#define X(name, cin, ks, md, dbg, var) Variant{name, cin, ks, md, make_conv_pp_launch<ConvPpCfg<cin, ks, md, dbg, var>>(name)},
      X("c128k11_pp", 128, 11, 5, 0, 0) X("c128k7_pp", 128, 7, 5, 0, 0) X("c256k11_pp", 256, 11, 5, 0, 0) X("c256k7_pp", 256, 7, 5, 0, 0)
      X("c128k11_pp_tl", 128, 11, 5, 1, 0)
      X("c128k11_pp_v1", 128, 11, 5, 0, 1) X("c128k11_pp_v2", 128, 11, 5, 0, 2)
      X("c128k11_pp_a16", 128, 11, 5, 0, 16) X("c128k11_pp_a32", 128, 11, 5, 0, 32) X("c128k11_pp_a64", 128, 11, 5, 0, 64)
      X("c128k11_pp_a112", 128, 11, 5, 0, 112)
#undef X
  };
  return v;
}

__global__ void fill_bf16_kernel(bf16_t* p, long long n, unsigned seed, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = (bf16_t)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
}

static long long* g_timeline = nullptr;  // device buffer the next evmi_bench_conv_tc call hands to the kernel
}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_bench_num_variants(void) { return (int)variants().size(); }
const char* evmi_bench_variant_name(int i) {
  return (i >= 0 && i < (int)variants().size()) ? variants()[i].name : "";
}

// Times `iters` launches of variant `name` as a (c -> c_out) convolution over [B, T] rows with
// dilation `dil`; c_out = 0 means c_out = c_in.  Returns mean milliseconds through *ms_out.
int evmi_bench_conv_tc(const char* name, int B, int T, int c_out, int dil, int with_residual, float pre_slope,
                       int iters, float* ms_out, double* flops_out) {
  const Variant* v = nullptr;
  for (const Variant& x : variants())
    if (!strcmp(x.name, name)) v = &x;
  if (!v) return fail(EVMI_ERR_INVALID_ARG, std::string("bench: unknown variant ") + name);
  const int cin = v->c_in, ks = v->ks;
  if (c_out <= 0) c_out = cin;
  if (c_out % v->launch.bm) return fail(EVMI_ERR_INVALID_ARG, "bench: c_out not a multiple of BM");
  const size_t xe = (size_t)B * T * cin, oe = (size_t)B * T * c_out, we = (size_t)c_out * ks * cin;
  bf16_t *x = nullptr, *w = nullptr, *out = nullptr, *res = nullptr;
  float* bias = nullptr;
  EVMI_HIP_CHECK(hipMalloc((void**)&x, xe * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&w, we * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&out, oe * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&res, oe * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&bias, (size_t)c_out * 4));
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((xe + 255) / 256)), dim3(256), 0, 0, x, (long long)xe, 1u, 1.f);
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((we + 255) / 256)), dim3(256), 0, 0, w, (long long)we, 2u, 0.05f);
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((oe + 255) / 256)), dim3(256), 0, 0, res, (long long)oe, 3u, 1.f);
  EVMI_HIP_CHECK(hipMemset(bias, 0, (size_t)c_out * 4));
  ConvTcArgs a;
  a.x = x; a.w = w; a.bias = bias; a.res = with_residual ? res : nullptr; a.out = out;
  a.t_in = T; a.n_rows = T; a.c_out = c_out; a.dil = dil; a.pad = dil * (ks - 1) / 2;
  a.x_batch_stride = (long long)T * cin; a.out_batch_stride = (long long)T * c_out;
  a.out_row_stride = c_out; a.out_shift = 0; a.out_limit = (long long)T * c_out;
  a.pre_slope = pre_slope; a.post_slope = 1.f; a.out_scale = 1.f; a.accumulate = 0;
  a.timeline = g_timeline;
  hipEvent_t e0, e1;
  EVMI_HIP_CHECK(hipEventCreate(&e0));
  EVMI_HIP_CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) {
    int rc = launch_conv_tc(&v->launch, a, B, 0);
    if (rc) return rc;
  }
  EVMI_HIP_CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) {
    int rc = launch_conv_tc(&v->launch, a, B, 0);
    if (rc) return rc;
  }
  EVMI_HIP_CHECK(hipEventRecord(e1, 0));
  EVMI_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  EVMI_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / iters;
  if (flops_out) *flops_out = 2.0 * B * (double)T * c_out * ks * cin;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(out); (void)hipFree(res); (void)hipFree(bias);
  return EVMI_OK;
}

}  // extern "C"

// s_memtime stamps (8 waves x 128) of one workgroup of a conv_tc debug variant (ABL bit 128), taken on the last of 3 launches.
extern "C" int evmi_bench_conv_tc_timeline(const char* name, int B, int T, int dil, int with_residual, float pre_slope,
                                           long long* stamps_host) {
  long long* tl = nullptr;
  EVMI_HIP_CHECK(hipMalloc((void**)&tl, 8 * 128 * 8));
  EVMI_HIP_CHECK(hipMemset(tl, 0, 8 * 128 * 8));
  g_timeline = tl;
  float ms = 0.f;
  double fl = 0.0;
  const int rc = evmi_bench_conv_tc(name, B, T, 0, dil, with_residual, pre_slope, 1, &ms, &fl);
  g_timeline = nullptr;
  if (rc) return rc;
  EVMI_HIP_CHECK(hipDeviceSynchronize());
  EVMI_HIP_CHECK(hipMemcpy(stamps_host, tl, 8 * 128 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(tl);
  return EVMI_OK;
}

// Cycle timeline of workgroup 0 of the fused pair kernel (debug instantiation): returns up to `cap`
// s_memtime stamps (tile start, x committed, one per step, loops done, stores issued, ... per tile).
extern "C" int evmi_debug_pair_timeline(int c, int ks, int dil, int B, int T, long long* stamps_host, int cap) {
  using namespace evmi;
  PairLaunch L;
  if (c == 64 && ks == 11) L = make_pair_launch<PairCfg<64, 11, 256, 2, 5, 8, 1, 2, 0, 1>>("dbg_pair_c64k11");
  else if (c == 64 && ks == 3) L = make_pair_launch<PairCfg<64, 3, 256, 2, 5, 8, 1>>("dbg_pair_c64k3");
  else if (c == 32 && ks == 11) L = make_pair_launch<PairCfg<32, 11, 512, 11, 5, 8, 1, 1>>("dbg_pair_c32k11");
  else if (c == 32 && ks == 3) L = make_pair_launch<PairCfg<32, 3, 512, 3, 5, 8, 1, 2>>("dbg_pair_c32k3");
  else return fail(EVMI_ERR_INVALID_ARG, "debug timeline: unsupported (c, ks)");
  const size_t xe = (size_t)B * T * c, we = (size_t)2 * c * c * ks;
  bf16_t *x = nullptr, *w = nullptr, *out = nullptr;
  float* bias = nullptr;
  long long* tl = nullptr;
  EVMI_HIP_CHECK(hipMalloc((void**)&x, xe * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&w, we * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&out, xe * 2));
  EVMI_HIP_CHECK(hipMalloc((void**)&bias, (size_t)2 * c * 4));
  EVMI_HIP_CHECK(hipMalloc((void**)&tl, 256 * 8));
  EVMI_HIP_CHECK(hipMemset(tl, 0, 256 * 8));
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((xe + 255) / 256)), dim3(256), 0, 0, x, (long long)xe, 1u, 1.f);
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((we + 255) / 256)), dim3(256), 0, 0, w, (long long)we, 2u, 0.05f);
  EVMI_HIP_CHECK(hipMemset(bias, 0, (size_t)2 * c * 4));
  PairArgs a;
  a.x = x; a.w1 = w; a.w2 = w + (size_t)c * c * ks; a.b1 = bias; a.b2 = bias + c; a.out = out;
  a.T = T; a.dil1 = dil; a.slope = 0.1f; a.post_slope = 1.f; a.out_scale = 1.f; a.accumulate = 0; a.timeline = tl;
  for (int i = 0; i < 2; ++i) {
    int rc = launch_resblock_pair(&L, a, B, 256, 0);
    if (rc) return rc;
  }
  EVMI_HIP_CHECK(hipDeviceSynchronize());
  EVMI_HIP_CHECK(hipMemcpy(stamps_host, tl, (size_t)(cap < 256 ? cap : 256) * 8, hipMemcpyDeviceToHost));
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(out); (void)hipFree(bias); (void)hipFree(tl);
  return EVMI_OK;
}
