/* Tuning aids exported by libevmi_hip.so next to the product ABI (include/evmi.h).
 * Not part of the drop-in boundary: used by tools/sweep_conv.py only. */
#ifndef EVMI_BENCH_H
#define EVMI_BENCH_H
#ifdef __cplusplus
extern "C" {
#endif
int evmi_bench_num_variants(void);
const char* evmi_bench_variant_name(int i);
/* Mean milliseconds of `iters` launches of conv_tc variant `name` on synthetic resident data. */
int evmi_bench_conv_tc(const char* name, int B, int T, int c_out, int dil, int with_residual,
                       float pre_slope, int iters, float* ms_out, double* flops_out);
/* s_memtime stamps of workgroup 0 of the fused residual-pair kernel (debug instantiation). */
int evmi_debug_pair_timeline(int c, int ks, int dil, int B, int T, long long* stamps_host, int cap);
#ifdef __cplusplus
}
#endif
#endif
