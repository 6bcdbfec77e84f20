// Multi-head self-attention for TRAINING (SURVEY.md 8a F2 / 8b evmi_mha_{fwd,bwd}): flash-style forward that keeps only the
// per-query log-sum-exp, and a backward that recomputes the probabilities from it -- no [B][T][T] tensor exists in either pass.
//
//   qkv [3D][B][T] channel-major (q rows, k rows, v rows; head h owns channels h*DH .. h*DH+DH-1), lens [B] (key padding mask)
//   forward : S = scale * Q K^T ; P = softmax_keys(S) ; Pd = dropout(P) ; O = Pd V           -> out [D][B][T], lse [B][H][T]
//   backward: D_q = <dO_q, O_q> ; dPd = dO V^T ; dS = P * (mask/(1-p) * dPd - D) ;
//             dQ = scale * dS K ; dK = scale * dS^T Q ; dV = Pd^T dO                          -> dqkv [3D][B][T]
//
// Attention dropout (torch.nn.MultiheadAttention applies it to the normalised probabilities) uses the counter-based generator
// of common.h (`uniform01`, the one evmi_dropout_f32 draws from): element ((b * T + q) * T + k) of the stream seeded with seed + h, so the backward regenerates the mask.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).  Both passes use the transposed-problem trick of
// the inference kernel (fs2_ops.hip): computing S^T = K Q^T leaves a score tile as lane = query, registers = keys, which is
// exactly the B-operand layout of the next product over the keys (step r contracts the two keys the half-waves hold in register
// r), so probabilities and score gradients never leave the registers.
//   * dq kernel : one wave owns 32 queries (Q^T and dO^T in registers), walks the key tiles (K, V staged in LDS).
//   * dkv kernel: one wave owns 32 keys (K^T and V^T in registers), walks the query tiles (Q, dO, lse, D staged in LDS).
// Each (query, key) tile is therefore recomputed twice in the backward; in exchange every output element has ONE writer and a
// fixed summation order: the gradients are bitwise reproducible (the checkpoint / resume test relies on that).
#include "common.h"

namespace evmi {

// value at p (p must be a valid address for every lane), zero for lanes that are not live.  The load is unconditional and the
// select follows it: a load under a per-lane condition is compiled as a branch around it with the memory counter drained behind
// every one (one dependent round trip per element: 128 of them in the prologue of a 128-wide head, 32 per key tile).
__device__ __forceinline__ float live_load(const float* __restrict__ p, bool live) {
  const float v = *p;
  return live ? v : 0.f;
}


// register r of a 32x32 accumulator <-> row index within the tile, for half-wave kh
__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// Dropout factors (keep or 0) of the 16 accumulator registers of a 32 x 32 score tile.
//   attn_drop_rows: the registers are keys k0 + acc_row(r, kh) of ONE query row (row_base = the row's first element): registers r, r + 1
//                   (r even) are the two elements of a pair -- one hash for both (pairs: T even and the tensor below 2^33 elements)
//   attn_drop_cols: the lane holds ONE key, the registers are queries q0 + acc_row(r, kh): the pair's other element sits in the
//                   neighbouring lane (key ^ 1) at the same register -- each lane hashes the registers of its own parity and the two
//                   exchange (one DPP move per hash)
// Same bits as uniform01(seed, element index) per element (common.h), which the other parities / sizes take.
__device__ __forceinline__ void attn_drop_rows(float (&dm)[16], bool pairs, unsigned long long seed, unsigned long long row_base, int k0, int kh,
                                               float p_drop, float keep) {
  if (pairs) {
    const unsigned jb = (unsigned)(row_base >> 1) + (unsigned)(k0 >> 1) + 2u * (unsigned)kh;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float u0, u1;
      dropout_pair(seed, jb + (unsigned)(((r & 3) >> 1) + 4 * (r >> 2)), u0, u1);
      dm[r] = u0 >= p_drop ? keep : 0.f;
      dm[r + 1] = u1 >= p_drop ? keep : 0.f;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) dm[r] = uniform01(seed, row_base + (unsigned long long)(k0 + acc_row(r, kh))) >= p_drop ? keep : 0.f;
  }
}
// (four registers at a time -- 4 g4 .. 4 g4 + 3 -- so that the factors do not stay live across the whole tile)
__device__ __forceinline__ void attn_drop_cols(float (&dm)[4], int g4, bool pairs, unsigned long long seed, unsigned long long batch_base, int tk,
                                               int tkc, int q0, int kh, int T, float p_drop, float keep) {
  if (pairs) {
    const unsigned par = (unsigned)tk & 1u, th = (unsigned)T >> 1;
    const unsigned jb = (unsigned)((batch_base + (unsigned long long)(tk & ~1)) >> 1);  // (the pair's index: not the clamped key's)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int tq = q0 + 8 * g4 + 4 * kh + 2 * c + (int)par;  // query of register 4 g4 + 2 c + par
      const unsigned mine = dropout_hash(seed, jb + (unsigned)min(tq, T - 1) * th, 0u);
      const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]: lane ^ 1
      const unsigned h0 = par ? other : mine, h1 = par ? mine : other;                              // registers 4 g4 + 2 c and + 1
      dm[2 * c] = dropout_u16(h0, par) >= p_drop ? keep : 0.f;
      dm[2 * c + 1] = dropout_u16(h1, par) >= p_drop ? keep : 0.f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int tq = q0 + 8 * g4 + 4 * kh + e;
      dm[e] = uniform01(seed, batch_base + (unsigned long long)tkc + (unsigned long long)min(tq, T - 1) * (unsigned long long)T) >= p_drop ? keep : 0.f;
    }
  }
}
static inline bool attn_drop_pairs(int B, int T) { return !(T & 1) && (unsigned long long)B * T * T < (1ull << 33); }

// A [DH][32] tile of a channel-major [DH][B][T] slice, columns t0 .. t0 + 31, by LDS-direct loads straight into the operand layout: rows of 32 floats, element (d, t) at d * 32 + (t ^ (d & 31)).
// The XOR swizzle replaces the odd row stride (a tile is read along its rows and across them: both are conflict-free), and is
// applied on the SOURCE side -- lane (d, p) of a request fetches column p ^ (d & 31) -- because the hardware writes lane i of a
// request to base + 4 i.  Columns past T are clamped to T - 1: finite values that only ever meet zero probabilities.  Two
// generations of every tile: step i + 1 lands while step i computes, one barrier per step.
typedef __attribute__((address_space(3))) float atf_lds_float_t;
typedef __attribute__((address_space(1))) const float atf_glb_float_t;
template <int DH>
__device__ __forceinline__ void tile_request_swz(float* __restrict__ dst, const float* __restrict__ src, long long N, int t0, int T, int tid) {
  const int d0 = tid >> 5, p = tid & 31;
  float* l = dst + (tid & ~63);
#pragma unroll
  for (int i = 0; i < DH / 8; ++i) {
    const int d = d0 + 8 * i;
    const float* g = src + (long long)d * N + min(t0 + (p ^ (d & 31)), T - 1);
    __builtin_amdgcn_global_load_lds((atf_glb_float_t*)g, (atf_lds_float_t*)(l + 256 * i), 4, 0, 0);
  }
}
// Per-lane offsets of the two kinds of read, computed once (32 registers; the rest of an address is an instruction immediate):
//   along a row : lane reads column ln of row 2 s + kh             -> (2 s + kh) * 32 + row_off[s & 15]
//   across rows : lane reads column acc_row(r, kh) of row 32 i + ln -> i * 1024 + col_off[r]
struct SwzOffsets {
  int row_off[16], col_off[16];
  __device__ __forceinline__ SwzOffsets(int ln, int kh) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      row_off[j] = ln ^ ((2 * j + kh) & 31);
      col_off[j] = ln * 32 + (((j & 3) + 8 * (j >> 2) + 4 * kh) ^ ln);
    }
  }
  __device__ __forceinline__ int along(int s, int kh) const { return (2 * s + kh) * 32 + row_off[s & 15]; }
  __device__ __forceinline__ int across(int i, int r) const { return i * 1024 + col_off[r]; }
};

// ---- forward -------------------------------------------------------------------------------------------------------------
// grid (ceil(T / 128), H, B), 256 threads: every wave owns 32 queries
template <int DH>
__global__ __launch_bounds__(256) void attention_train_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                 float* __restrict__ out, float* __restrict__ lse, int B, int T, int D,
                                                                 float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  extern __shared__ __attribute__((aligned(16))) float atf_lds[];
  float* Kb = atf_lds;                 // [2 generations][DH][32], swizzled
  float* Vb = atf_lds + 2 * DH * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const SwzOffsets sw(ln, kh);
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* q = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  if (len > 0) {
    tile_request_swz<DH>(Kb, kg, N, 0, T, tid);
    tile_request_swz<DH>(Vb, vg, N, 0, T, tid);
  }
  const int tq = blockIdx.x * 128 + wave * 32 + ln;
  const bool qlive = tq < T;
  const int tqc = min(tq, T - 1);  // clamped: loads are unconditional, dead lanes are zeroed by a select
  float qreg[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) qreg[s] = live_load(q + (long long)(2 * s + kh) * N + tqc, qlive) * scale;
  f32x16 acc[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const unsigned long long row_base = ((unsigned long long)b * T + (unsigned long long)(qlive ? tq : 0)) * T;

  for (int k0 = 0, gen = 0; k0 < len; k0 += 32, gen ^= 1) {
    lds_dma_barrier();  // this step's tiles have landed (explicit vmcnt(0) of every wave); the other generation's readers are done
    if (k0 + 32 < len) {
      tile_request_swz<DH>(Kb + (gen ^ 1) * DH * 32, kg, N, k0 + 32, T, tid);
      tile_request_swz<DH>(Vb + (gen ^ 1) * DH * 32, vg, N, k0 + 32, T, tid);
    }
    const float* Ks = Kb + gen * DH * 32;
    const float* Vs = Vb + gen * DH * 32;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[sw.along(s, kh)], qreg[s], st, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (k0 + acc_row(r, kh) >= len) st[r] = -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);  // finite: every processed tile has a valid key
    const float corr = expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_new);
      ps += st[r];  // the normaliser sums the probabilities BEFORE dropout
      if (p_drop > 0.f)
        st[r] = uniform01(seed + h, row_base + (unsigned long long)(k0 + acc_row(r, kh))) >= p_drop ? st[r] * keep : 0.f;
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * corr + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[sw.across(i, r)], st[r], acc[i], 0, 0, 0);
    }
  }
  if (!qlive) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;  // an item of length 0 has no keys: zero rows
  float* o = out + ((long long)(h * DH) * B + b) * T + tq;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)(i * 32 + acc_row(r, kh)) * N] = acc[i][r] * inv;
  if (kh == 0) lse[((long long)b * H + h) * T + tq] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;  // +inf: exp(s - lse) = 0
}

// ---- backward, step 0: D[b][h][t] = sum_d dO[d][b][t] * O[d][b][t] ---------------------------------------------------------
__global__ void attention_rowdot_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ dsum, int B, int T,
                                        int H, int DH) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * T) return;
  const int t = (int)(i % T), h = (int)((i / T) % H), b = (int)(i / ((long long)T * H));
  const long long N = (long long)B * T;
  const long long base = ((long long)(h * DH) * B + b) * T + t;
  float acc = 0.f;
  int d = 0;
  for (; d + 8 <= DH; d += 8) {  // sixteen loads in flight (a plain loop is one memory round trip per d: 53 us at DH = 128); same fmaf order
    float ov[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ov[u] = o[base + (d + u) * N]; dv[u] = d_o[base + (d + u) * N]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(ov[u], dv[u], acc);
  }
  for (; d < DH; ++d) acc = fmaf(o[base + d * N], d_o[base + d * N], acc);
  dsum[i] = acc;
}

// ---- backward, dQ: grid (ceil(T / 128), H, B); every wave owns 32 queries and walks the key tiles -----------------------------
template <int DH>
__global__ __launch_bounds__(256) void attention_train_dq_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                const float* __restrict__ dsum, float* __restrict__ dqkv, int B, int T,
                                                                int D, float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  extern __shared__ __attribute__((aligned(16))) float atf_lds[];
  float* Kb = atf_lds;                 // [2 generations][DH][32], swizzled
  float* Vb = atf_lds + 2 * DH * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const SwzOffsets sw(ln, kh);
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* q = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  const float* dog = d_o + ((long long)(h * DH) * B + b) * T;
  if (len > 0) {
    tile_request_swz<DH>(Kb, kg, N, 0, T, tid);
    tile_request_swz<DH>(Vb, vg, N, 0, T, tid);
  }
  const int tq = blockIdx.x * 128 + wave * 32 + ln;
  const bool qlive = tq < T;
  const int tqc = min(tq, T - 1);  // clamped: loads are unconditional, dead lanes are zeroed by a select
  float qreg[DH / 2], doreg[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) {
    qreg[s] = live_load(q + (long long)(2 * s + kh) * N + tqc, qlive) * scale;
    doreg[s] = live_load(dog + (long long)(2 * s + kh) * N + tqc, qlive);
  }
  const float my_lse_raw = lse[((long long)b * H + h) * T + tqc];
  const float my_lse = qlive ? my_lse_raw : INFINITY;
  const float my_d = live_load(dsum + ((long long)b * H + h) * T + tqc, qlive);
  const float keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const unsigned long long row_base = ((unsigned long long)b * T + (unsigned long long)(qlive ? tq : 0)) * T;
  f32x16 acc[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int k0 = 0, gen = 0; k0 < len; k0 += 32, gen ^= 1) {
    lds_dma_barrier();
    if (k0 + 32 < len) {
      tile_request_swz<DH>(Kb + (gen ^ 1) * DH * 32, kg, N, k0 + 32, T, tid);
      tile_request_swz<DH>(Vb + (gen ^ 1) * DH * 32, vg, N, k0 + 32, T, tid);
    }
    const float* Ks = Kb + gen * DH * 32;
    const float* Vs = Vb + gen * DH * 32;
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[sw.along(s, kh)], qreg[s], st, 0, 0, 0);   // S^T  = K Q^T
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[sw.along(s, kh)], doreg[s], dp, 0, 0, 0);  // dPd^T = V dO^T
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, kh);
      const float pr = key < len ? expf(st[r] - my_lse) : 0.f;
      float g = dp[r];
      if (p_drop > 0.f) g = uniform01(seed + h, row_base + (unsigned long long)key) >= p_drop ? g * keep : 0.f;
      st[r] = pr * (g - my_d);  // dS^T as it lies: lane = query, register = key
    }
#pragma unroll
    for (int i = 0; i < DH / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)  // dQ^T [d][query] += K^T [d][key] dS^T [key][query]
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[sw.across(i, r)], st[r], acc[i], 0, 0, 0);
  }
  if (!qlive) return;
  float* o = dqkv + ((long long)(h * DH) * B + b) * T + tq;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)(i * 32 + acc_row(r, kh)) * N] = acc[i][r] * scale;
}

// ---- backward, dK and dV: grid (ceil(T / 128), H, B); every wave owns 32 keys and walks the query tiles ---------------------
template <int DH>
__global__ __launch_bounds__(256) void attention_train_dkv_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                 const float* __restrict__ dsum, float* __restrict__ dqkv, int B, int T,
                                                                 int D, float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  extern __shared__ __attribute__((aligned(16))) float atf_lds[];
  float* Qb = atf_lds;                 // [2 generations][DH][32], swizzled
  float* Ob = atf_lds + 2 * DH * 32;
  float* stat = atf_lds + 4 * DH * 32; // [2 generations][lse of the 32 queries | D of the 32 queries]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const SwzOffsets sw(ln, kh);
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* qg = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  const float* dog = d_o + ((long long)(h * DH) * B + b) * T;
  const int tk = blockIdx.x * 128 + wave * 32 + ln;
  const bool klive = tk < len;
  const int tkc = min(tk, T - 1);  // padded keys receive no probability mass: zero gradients
  float kreg[DH / 2], vreg[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) {
    kreg[s] = live_load(kg + (long long)(2 * s + kh) * N + tkc, klive) * scale;
    vreg[s] = live_load(vg + (long long)(2 * s + kh) * N + tkc, klive);
  }
  const float keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  f32x16 acck[DH / 32], accv[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acck[i][r] = accv[i][r] = 0.f;
  const bool block_live = blockIdx.x * 128 < len;  // uniform per workgroup
  // wave 0 also brings the tile's 32 (lse, D) pairs: lanes 0-31 the lse, lanes 32-63 D, clamped (queries past T are masked below)
  const float* stat_src = (kh ? dsum : lse) + ((long long)b * H + h) * T;
  auto request = [&](int q0, int gen) {
    tile_request_swz<DH>(Qb + gen * DH * 32, qg, N, q0, T, tid);
    tile_request_swz<DH>(Ob + gen * DH * 32, dog, N, q0, T, tid);
    if (wave == 0) __builtin_amdgcn_global_load_lds((atf_glb_float_t*)(stat_src + min(q0 + ln, T - 1)), (atf_lds_float_t*)(stat + 64 * gen), 4, 0, 0);
  };
  if (block_live) request(0, 0);

  for (int q0 = 0, gen = 0; q0 < T && block_live; q0 += 32, gen ^= 1) {
    lds_dma_barrier();
    if (q0 + 32 < T) request(q0 + 32, gen ^ 1);
    const float* Qs = Qb + gen * DH * 32;
    const float* Os = Ob + gen * DH * 32;
    const float* lse_s = stat + 64 * gen;
    const float* d_s = lse_s + 32;
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[sw.along(s, kh)], kreg[s], st, 0, 0, 0);  // S   = Q K^T : lane = key, registers = queries
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[sw.along(s, kh)], vreg[s], dp, 0, 0, 0);  // dPd = dO V^T
    }
    f32x16 pd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, kh);
      const int tq = q0 + qi;
      float pr = expf(st[r] - lse_s[qi]);
      pr = (klive && tq < T) ? pr : 0.f;
      float mk = 1.f;
      if (p_drop > 0.f) mk = uniform01(seed + h, ((unsigned long long)b * T + (unsigned long long)min(tq, T - 1)) * T + (unsigned long long)min(tk, T - 1)) >= p_drop ? keep : 0.f;
      pd[r] = pr * mk;                        // Pd   : the dropped-out probabilities that multiplied V
      st[r] = pr * (dp[r] * mk - d_s[qi]);    // dS
    }
#pragma unroll
    for (int i = 0; i < DH / 32; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // (column acc_row(r, kh) of the tile: the query of register r)
        accv[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[sw.across(i, r)], pd[r], accv[i], 0, 0, 0);  // dV^T += dO^T Pd
        acck[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[sw.across(i, r)], st[r], acck[i], 0, 0, 0);  // dK^T += Q^T dS
      }
  }
  if (tk >= T) return;
  float* dk = dqkv + ((long long)(D + h * DH) * B + b) * T + tk;
  float* dv = dqkv + ((long long)(2 * D + h * DH) * B + b) * T + tk;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk[(long long)(i * 32 + acc_row(r, kh)) * N] = acck[i][r] * scale;
      dv[(long long)(i * 32 + acc_row(r, kh)) * N] = accv[i][r];
    }
}


// ==== bf16-operand variants (precision = "bf16": what BASELINE config 3 names) =================================================
// Same structure on v_mfma_f32_32x32x16_bf16: Q / K / V / dO, the probabilities and the score gradients are rounded to bf16 on
// their way into the matrix cores; scores, softmax statistics, D and every accumulator stay fp32.  A contraction over the head
// dimension reads 8 consecutive channels per lane, so its LDS operand is staged [position][channel]; a contraction over the keys
// (queries) of a tile takes its k slots in the order the half-waves already hold the score tile's registers -- slot (half, e) of
// block kb is position 16 kb + 8 (e >> 2) + 4 half + (e & 3) -- so P^T / dS^T feed it straight from registers and the other
// operand is staged [channel][position] (two 8-byte reads).  Tiles that take part in both kinds of product are staged twice.
//
// The walked tiles arrive by LDS-direct loads (global_load_lds_dword): the fp32 [DH][32] tile of step i + 1 is requested into a
// raw staging area right after step i's operands are in place and lands under step i's matrix work without holding a register;
// at the top of step i + 1 every thread converts the 16 elements IT requested (no cross-thread hazard on the raw area) into the
// bf16 operand layouts.  (Prefetching through registers made the compiler park the 32 in-flight values in accumulation
// registers with a full vmcnt(0) wait behind EVERY load: 32 serial round trips per tile in the dK/dV kernel.)
// The element loops are straight-line: the probability is always computed and masked by a select, the per-query statistics of
// the dK/dV kernel are read as 16-byte LDS vectors, dropout is a template parameter (its generator costs more than the exp).
constexpr int ATB_PD = 8;  // bf16 row padding

typedef __attribute__((address_space(3))) float at_lds_float_t;
typedef __attribute__((address_space(1))) const float at_glb_float_t;

// request the [DH][32] fp32 tile of a channel-major slice (columns t0 .. t0 + 31, clamped to T - 1) into raw[DH * 32]: element
// v = tid + 256 i (channel v >> 5, column v & 31) lands at raw[v]
template <int DH>
__device__ __forceinline__ void tile_request(float* __restrict__ raw, const float* __restrict__ src, long long N, int t0, int T, int tid) {
  const float* g = src + (long long)(tid >> 5) * N + min(t0 + (tid & 31), T - 1);
  float* l = raw + (tid & ~63);
#pragma unroll
  for (int i = 0; i < DH / 8; ++i)
    __builtin_amdgcn_global_load_lds((at_glb_float_t*)(g + (long long)(8 * i) * N), (at_lds_float_t*)(l + 256 * i), 4, 0, 0);
}
// the thread's own 16 elements of a landed tile -> bf16 [position][channel] (PC) and / or [channel][position] (CP), zero past T
template <int DH, bool PC, bool CP>
__device__ __forceinline__ void tile_convert(const float* __restrict__ raw, bf16_t* __restrict__ x_pc, bf16_t* __restrict__ x_cp, int t0, int T,
                                             int tid) {
  constexpr int LP = DH + ATB_PD, LC = 32 + ATB_PD;
  const int tt = tid & 31, d0 = tid >> 5;
  const bool in = t0 + tt < T;
  float r[DH / 8];
#pragma unroll
  for (int i = 0; i < DH / 8; ++i) r[i] = raw[tid + 256 * i];
#pragma unroll
  for (int i = 0; i < DH / 8; ++i) {
    const bf16_t val = (bf16_t)(in ? r[i] : 0.f);
    if (PC) x_pc[tt * LP + d0 + 8 * i] = val;
    if (CP) x_cp[(d0 + 8 * i) * LC + tt] = val;
  }
}

// A operand of a contraction over the tile's positions: lane (channel row, half) takes positions 16 kb + 4 half + {0..3} and + 8
__device__ __forceinline__ bf16x8 load_pos_slots(const bf16_t* __restrict__ row, int kb, int kh) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row + 16 * kb + 4 * kh);
  const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row + 16 * kb + 4 * kh + 8);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int DH, int DROP>  // DROP 0: no dropout; 1: one hash per element; 2: one hash per pair of elements (attn_drop_pairs)
__global__ __launch_bounds__(256, 2) void attention_train_fwd_bf16_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                      float* __restrict__ out, float* __restrict__ lse, int B, int T, int D,
                                                                      float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  constexpr int LP = DH + ATB_PD, LC = 32 + ATB_PD;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * LP];   // [key][channel]
  __shared__ __attribute__((aligned(16))) bf16_t Vs[DH * LC];   // [channel][key]
  __shared__ __attribute__((aligned(16))) float rawK[DH * 32], rawV[DH * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* q = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  if (len > 0) {
    tile_request<DH>(rawK, kg, N, 0, T, tid);
    tile_request<DH>(rawV, vg, N, 0, T, tid);
  }
  const int tq = blockIdx.x * 128 + wave * 32 + ln;
  const bool qlive = tq < T;
  const int tqc = min(tq, T - 1);  // clamped: loads are unconditional, dead lanes are zeroed by a select
  bf16x8 qreg[DH / 16];
#pragma unroll
  for (int s = 0; s < DH / 16; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) qreg[s][e] = (bf16_t)(live_load(q + (long long)(16 * s + 8 * kh + e) * N + tqc, qlive) * scale);
  f32x16 acc[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float keep = DROP ? 1.f / (1.f - p_drop) : 1.f;
  constexpr bool drop_pairs = DROP == 2;
  const unsigned long long row_base = ((unsigned long long)b * T + (unsigned long long)(qlive ? tq : 0)) * T;

  for (int k0 = 0; k0 < len; k0 += 32) {
    lds_dma_barrier();  // the requested tile has landed (vmcnt(0) of every wave); the previous step's operand reads are over
    tile_convert<DH, true, false>(rawK, Ks, nullptr, k0, T, tid);
    tile_convert<DH, false, true>(rawV, nullptr, Vs, k0, T, tid);
    __syncthreads();
    if (k0 + 32 < len) {
      tile_request<DH>(rawK, kg, N, k0 + 32, T, tid);
      tile_request<DH>(rawV, vg, N, k0 + 32, T, tid);
    }
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&Ks[ln * LP + 16 * s + 8 * kh]), qreg[s], st, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = k0 + acc_row(r, kh) < len ? st[r] : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);  // finite: key k0 of the tile is live
    const float corr = __expf(m_run - m_new);
    float ps = 0.f;
    bf16x8 pb[2];
    float dm[16];
    if (DROP) attn_drop_rows(dm, drop_pairs, seed + h, row_base, k0, kh, p_drop, keep);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pr = __expf(st[r] - m_new);
      ps += pr;
      if (DROP) pr = dm[r] != 0.f ? pr * keep : 0.f;
      pb[r >> 3][r & 7] = (bf16_t)pr;
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * corr + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] *= corr;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_pos_slots(&Vs[(i * 32 + ln) * LC], kb, kh), pb[kb], acc[i], 0, 0, 0);
    }
  }
  if (!qlive) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  float* o = out + ((long long)(h * DH) * B + b) * T + tq;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)(i * 32 + acc_row(r, kh)) * N] = acc[i][r] * inv;
  if (kh == 0 && lse) lse[((long long)b * H + h) * T + tq] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;  // (inference: no lse)
}

template <int DH, int DROP>  // DROP 0: no dropout; 1: one hash per element; 2: one hash per pair of elements (attn_drop_pairs)
__global__ __launch_bounds__(256, 2) void attention_train_dq_bf16_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                     const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                     const float* __restrict__ dsum, float* __restrict__ dqkv, int B, int T,
                                                                     int D, float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  constexpr int LP = DH + ATB_PD, LC = 32 + ATB_PD;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * LP];   // [key][channel]: S^T = K Q^T
  __shared__ __attribute__((aligned(16))) bf16_t Kt[DH * LC];   // [channel][key]: dQ^T += K^T dS^T
  __shared__ __attribute__((aligned(16))) bf16_t Vs[32 * LP];   // [key][channel]: dPd^T = V dO^T
  __shared__ __attribute__((aligned(16))) float rawK[DH * 32], rawV[DH * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* q = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  const float* dog = d_o + ((long long)(h * DH) * B + b) * T;
  if (len > 0) {
    tile_request<DH>(rawK, kg, N, 0, T, tid);
    tile_request<DH>(rawV, vg, N, 0, T, tid);
  }
  const int tq = blockIdx.x * 128 + wave * 32 + ln;
  const bool qlive = tq < T;
  const int tqc = min(tq, T - 1);  // clamped: loads are unconditional, dead lanes are zeroed by a select
  bf16x8 qreg[DH / 16], doreg[DH / 16];
#pragma unroll
  for (int s = 0; s < DH / 16; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long long off = (long long)(16 * s + 8 * kh + e) * N + tqc;
      qreg[s][e] = (bf16_t)(live_load(q + off, qlive) * scale);
      doreg[s][e] = (bf16_t)live_load(dog + off, qlive);
    }
  const float my_lse_raw = lse[((long long)b * H + h) * T + tqc];
  const float my_lse = qlive ? my_lse_raw : INFINITY;
  const float my_d = live_load(dsum + ((long long)b * H + h) * T + tqc, qlive);
  const float keep = DROP ? 1.f / (1.f - p_drop) : 1.f;
  constexpr bool drop_pairs = DROP == 2;
  const unsigned long long row_base = ((unsigned long long)b * T + (unsigned long long)(qlive ? tq : 0)) * T;
  f32x16 acc[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int k0 = 0; k0 < len; k0 += 32) {
    lds_dma_barrier();
    tile_convert<DH, true, true>(rawK, Ks, Kt, k0, T, tid);
    tile_convert<DH, true, false>(rawV, Vs, nullptr, k0, T, tid);
    __syncthreads();
    if (k0 + 32 < len) {
      tile_request<DH>(rawK, kg, N, k0 + 32, T, tid);
      tile_request<DH>(rawV, vg, N, k0 + 32, T, tid);
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&Ks[ln * LP + 16 * s + 8 * kh]), qreg[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&Vs[ln * LP + 16 * s + 8 * kh]), doreg[s], dp, 0, 0, 0);
    }
    bf16x8 dsb[2];
    float dm[16];
    if (DROP) attn_drop_rows(dm, drop_pairs, seed + h, row_base, k0, kh, p_drop, keep);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, kh);
      float pr = __expf(st[r] - my_lse);
      pr = key < len ? pr : 0.f;
      float g = dp[r];
      if (DROP) g = dm[r] != 0.f ? g * keep : 0.f;
      dsb[r >> 3][r & 7] = (bf16_t)(pr * (g - my_d));
    }
#pragma unroll
    for (int i = 0; i < DH / 32; ++i)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_pos_slots(&Kt[(i * 32 + ln) * LC], kb, kh), dsb[kb], acc[i], 0, 0, 0);
  }
  if (!qlive) return;
  float* o = dqkv + ((long long)(h * DH) * B + b) * T + tq;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)(i * 32 + acc_row(r, kh)) * N] = acc[i][r] * scale;
}

// dynamic LDS of the dK/dV kernel: four bf16 operand tiles, two raw fp32 tiles, two generations of the 32 queries' (lse, D)
template <int DH>
constexpr size_t attention_dkv_bf16_lds() {
  return (size_t)(2 * 32 * (DH + ATB_PD) + 2 * DH * (32 + ATB_PD)) * sizeof(bf16_t) + (size_t)(2 * DH * 32 + 2 * 64) * sizeof(float);
}

template <int DH, int DROP>  // DROP 0: no dropout; 1: one hash per element; 2: one hash per pair of elements (attn_drop_pairs)
__global__ __launch_bounds__(256, 2) void attention_train_dkv_bf16_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                                      const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                      const float* __restrict__ dsum, float* __restrict__ dqkv, int B, int T,
                                                                      int D, float scale, float p_drop, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  constexpr int LP = DH + ATB_PD, LC = 32 + ATB_PD;
  extern __shared__ __attribute__((aligned(16))) unsigned char at_dyn_lds[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(at_dyn_lds);  // [query][channel]
  bf16_t* Qt = Qs + 32 * LP;                           // [channel][query]
  bf16_t* Os = Qt + DH * LC;                           // dO [query][channel]
  bf16_t* Ot = Os + 32 * LP;                           // dO [channel][query]
  float* rawQ = reinterpret_cast<float*>(Ot + DH * LC);
  float* rawO = rawQ + DH * 32;
  float* stat = rawO + DH * 32;                        // [2 generations][lse of the 32 queries | D of the 32 queries]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* qg = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  const float* dog = d_o + ((long long)(h * DH) * B + b) * T;
  const bool block_live = blockIdx.x * 128 < len;
  // wave 0 also brings the tile's 32 (lse, D) pairs: lanes 0-31 the lse, lanes 32-63 D, clamped (queries past T are masked below)
  const float* stat_src = (kh ? dsum : lse) + ((long long)b * H + h) * T;
  if (block_live) {
    tile_request<DH>(rawQ, qg, N, 0, T, tid);
    tile_request<DH>(rawO, dog, N, 0, T, tid);
    if (wave == 0) __builtin_amdgcn_global_load_lds((at_glb_float_t*)(stat_src + min(ln, T - 1)), (at_lds_float_t*)stat, 4, 0, 0);
  }
  const int tk = blockIdx.x * 128 + wave * 32 + ln;
  const bool klive = tk < len;
  const int tkc = min(tk, T - 1);
  bf16x8 kreg[DH / 16], vreg[DH / 16];
#pragma unroll
  for (int s = 0; s < DH / 16; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long long off = (long long)(16 * s + 8 * kh + e) * N + tkc;
      kreg[s][e] = (bf16_t)(live_load(kg + off, klive) * scale);
      vreg[s][e] = (bf16_t)live_load(vg + off, klive);
    }
  const float keep = DROP ? 1.f / (1.f - p_drop) : 1.f;
  const unsigned long long batch_base = (unsigned long long)b * T * T;
  constexpr bool drop_pairs = DROP == 2;
  f32x16 acck[DH / 32], accv[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acck[i][r] = accv[i][r] = 0.f;

  for (int q0 = 0, gen = 0; q0 < T && block_live; q0 += 32, gen ^= 1) {
    lds_dma_barrier();
    tile_convert<DH, true, true>(rawQ, Qs, Qt, q0, T, tid);
    tile_convert<DH, true, true>(rawO, Os, Ot, q0, T, tid);
    __syncthreads();
    if (q0 + 32 < T) {
      tile_request<DH>(rawQ, qg, N, q0 + 32, T, tid);
      tile_request<DH>(rawO, dog, N, q0 + 32, T, tid);
      if (wave == 0)
        __builtin_amdgcn_global_load_lds((at_glb_float_t*)(stat_src + min(q0 + 32 + ln, T - 1)), (at_lds_float_t*)(stat + 64 * (gen ^ 1)), 4, 0, 0);
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&Qs[ln * LP + 16 * s + 8 * kh]), kreg[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&Os[ln * LP + 16 * s + 8 * kh]), vreg[s], dp, 0, 0, 0);
    }
    // the 16 queries of this half-wave's registers: (r & 3) + 8 (r >> 2) + 4 kh -> four 16-byte vectors of each statistic
    bf16x8 pdb[2], dsb[2];
    const float* st_l = stat + 64 * gen + 4 * kh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 lse_q = *reinterpret_cast<const f32x4*>(st_l + 8 * g4);
      const f32x4 d_q = *reinterpret_cast<const f32x4*>(st_l + 32 + 8 * g4);
      float dm[4];
      if (DROP) attn_drop_cols(dm, g4, drop_pairs, seed + h, batch_base, tk, tkc, q0, kh, T, p_drop, keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const int tq = q0 + 8 * g4 + 4 * kh + e;
        float pr = __expf(st[r] - lse_q[e]);
        pr = (klive && tq < T) ? pr : 0.f;
        const float mk = DROP ? dm[e] : 1.f;
        pdb[r >> 3][r & 7] = (bf16_t)(pr * mk);
        dsb[r >> 3][r & 7] = (bf16_t)(pr * (dp[r] * mk - d_q[e]));
      }
    }
#pragma unroll
    for (int i = 0; i < DH / 32; ++i)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        accv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_pos_slots(&Ot[(i * 32 + ln) * LC], kb, kh), pdb[kb], accv[i], 0, 0, 0);
        acck[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_pos_slots(&Qt[(i * 32 + ln) * LC], kb, kh), dsb[kb], acck[i], 0, 0, 0);
      }
  }
  if (tk >= T) return;
  float* dk = dqkv + ((long long)(D + h * DH) * B + b) * T + tk;
  float* dv = dqkv + ((long long)(2 * D + h * DH) * B + b) * T + tk;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk[(long long)(i * 32 + acc_row(r, kh)) * N] = acck[i][r] * scale;
      dv[(long long)(i * 32 + acc_row(r, kh)) * N] = accv[i][r];
    }
}

// The bf16 forward without dropout; lse may be NULL (the inference forward of fs2_ops.hip: evmi_attention_cbt_bf16 runs on this
// kernel too -- 149 -> 82 us per decoder layer against the register-staged inference kernel it replaced).
int launch_mha_fwd_bf16_plain(const float* qkv, const int* lens, float* out, float* lse_or_null, int B, int T, int D, int heads,
                              hipStream_t s) {
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  const dim3 grid((T + 127) / 128, heads, B);
  if (dh == 128) hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<128, 0>), grid, dim3(256), 0, s, qkv, lens, out, lse_or_null, B, T, D, scale, 0.f, SeedArg{0ull, nullptr});
  else if (dh == 64) hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<64, 0>), grid, dim3(256), 0, s, qkv, lens, out, lse_or_null, B, T, D, scale, 0.f, SeedArg{0ull, nullptr});
  else if (dh == 32) hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<32, 0>), grid, dim3(256), 0, s, qkv, lens, out, lse_or_null, B, T, D, scale, 0.f, SeedArg{0ull, nullptr});
  else return 1;
  return 0;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_mha_fwd_f32(const float* qkv_dev, const int* lens_dev, float* out_dev, float* lse_dev, int B, int T, int D, int heads,
                     float p_drop, unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!qkv_dev || !lens_dev || !out_dev || !lse_dev) return fail(EVMI_ERR_INVALID_ARG, "mha_fwd: null pointer");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads || p_drop < 0.f || p_drop >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "mha_fwd: shape / dropout");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "mha_fwd: grid limits");
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  const dim3 grid((T + 127) / 128, heads, B);
  hipStream_t s = (hipStream_t)stream;
#define EVMI_MHA_FWD(DH, SLOT)                                                                                                    \
  {                                                                                                                               \
    constexpr size_t lds = (size_t)4 * DH * 32 * sizeof(float);                                                                   \
    static thread_local bool configured_dev[kMaxDevices][3] = {};                                                \
    bool* configured = configured_dev[device_slot()];          \
    if (!configured[SLOT]) {                                                                                                      \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)attention_train_fwd_kernel<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds));                                                                              \
      configured[SLOT] = true;                                                                                                    \
    }                                                                                                                             \
    hipLaunchKernelGGL(attention_train_fwd_kernel<DH>, grid, dim3(256), lds, s, qkv_dev, lens_dev, out_dev, lse_dev, B, T, D,     \
                       scale, p_drop, seed);                                                                                      \
  }
  if (dh == 128) EVMI_MHA_FWD(128, 0)
  else if (dh == 64) EVMI_MHA_FWD(64, 1)
  else if (dh == 32) EVMI_MHA_FWD(32, 2)
  else return fail(EVMI_ERR_UNSUPPORTED, "mha_fwd: head dimension must be 32, 64 or 128");
#undef EVMI_MHA_FWD
  EVMI_LAUNCH_CHECK("mha_fwd");
  return EVMI_OK;
}

int evmi_mha_fwd_bf16(const float* qkv_dev, const int* lens_dev, float* out_dev, float* lse_dev, int B, int T, int D, int heads,
                      float p_drop, unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!qkv_dev || !lens_dev || !out_dev || !lse_dev) return fail(EVMI_ERR_INVALID_ARG, "mha_fwd_bf16: null pointer");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads || p_drop < 0.f || p_drop >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "mha_fwd_bf16: shape / dropout");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "mha_fwd_bf16: grid limits");
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  const dim3 grid((T + 127) / 128, heads, B);
  hipStream_t s = (hipStream_t)stream;
#define EVMI_MHA_FWD(DH)                                                                                                          \
  {                                                                                                                               \
    if (p_drop > 0.f && attn_drop_pairs(B, T))                                                                                    \
      hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<DH, 2>), grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, lse_dev, B, T, D, \
                         scale, p_drop, seed);                                                                                    \
    else if (p_drop > 0.f)                                                                                                        \
      hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<DH, 1>), grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, lse_dev, B, T, D, \
                         scale, p_drop, seed);                                                                                    \
    else                                                                                                                          \
      hipLaunchKernelGGL((attention_train_fwd_bf16_kernel<DH, 0>), grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, lse_dev, B, T, D, \
                         scale, p_drop, seed);                                                                                    \
  }
  if (dh == 128) EVMI_MHA_FWD(128)
  else if (dh == 64) EVMI_MHA_FWD(64)
  else if (dh == 32) EVMI_MHA_FWD(32)
  else return fail(EVMI_ERR_UNSUPPORTED, "mha_fwd_bf16: head dimension must be 32, 64 or 128");
#undef EVMI_MHA_FWD
  EVMI_LAUNCH_CHECK("mha_fwd_bf16");
  return EVMI_OK;
}

int evmi_mha_bwd_bf16(const float* qkv_dev, const int* lens_dev, const float* out_dev, const float* dout_dev, const float* lse_dev,
                      float* dsum_dev, float* dqkv_dev, int B, int T, int D, int heads, float p_drop, unsigned long long seed_value, const unsigned long long* seed_base_dev,
                      void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!qkv_dev || !lens_dev || !out_dev || !dout_dev || !lse_dev || !dsum_dev || !dqkv_dev) return fail(EVMI_ERR_INVALID_ARG, "mha_bwd_bf16: null pointer");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads || p_drop < 0.f || p_drop >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "mha_bwd_bf16: shape / dropout");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "mha_bwd_bf16: grid limits");
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * heads * T;
  hipLaunchKernelGGL(attention_rowdot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out_dev, dout_dev, dsum_dev, B, T, heads, dh);
  const dim3 grid((T + 127) / 128, heads, B);
#define EVMI_MHA_BWD_DROP(DH, DROP, SLOT)                                                                                        \
  {                                                                                                                               \
    constexpr size_t lds = attention_dkv_bf16_lds<DH>();                                                                          \
    static thread_local bool configured_dev[kMaxDevices][12] = {};                                               \
    bool* configured = configured_dev[device_slot()];          \
    if (!configured[SLOT]) {                                                                                                      \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)attention_train_dkv_bf16_kernel<DH, DROP>,                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                  \
      configured[SLOT] = true;                                                                                                    \
    }                                                                                                                             \
    hipLaunchKernelGGL((attention_train_dq_bf16_kernel<DH, DROP>), grid, dim3(256), 0, s, qkv_dev, lens_dev, dout_dev, lse_dev,   \
                       dsum_dev, dqkv_dev, B, T, D, scale, p_drop, seed);                                                         \
    hipLaunchKernelGGL((attention_train_dkv_bf16_kernel<DH, DROP>), grid, dim3(256), lds, s, qkv_dev, lens_dev, dout_dev, lse_dev,\
                       dsum_dev, dqkv_dev, B, T, D, scale, p_drop, seed);                                                         \
  }
#define EVMI_MHA_BWD(DH, SLOT)                                                                                                    \
  {                                                                                                                               \
    if (p_drop > 0.f && attn_drop_pairs(B, T)) EVMI_MHA_BWD_DROP(DH, 2, SLOT)                                                     \
    else if (p_drop > 0.f) EVMI_MHA_BWD_DROP(DH, 1, SLOT + 1)                                                                     \
    else EVMI_MHA_BWD_DROP(DH, 0, SLOT + 2)                                                                                       \
  }
  if (dh == 128) EVMI_MHA_BWD(128, 0)
  else if (dh == 64) EVMI_MHA_BWD(64, 3)
  else if (dh == 32) EVMI_MHA_BWD(32, 6)
  else return fail(EVMI_ERR_UNSUPPORTED, "mha_bwd_bf16: head dimension must be 32, 64 or 128");
#undef EVMI_MHA_BWD
#undef EVMI_MHA_BWD_DROP
  EVMI_LAUNCH_CHECK("mha_bwd_bf16");
  return EVMI_OK;
}

int evmi_mha_bwd_f32(const float* qkv_dev, const int* lens_dev, const float* out_dev, const float* dout_dev, const float* lse_dev,
                     float* dsum_dev, float* dqkv_dev, int B, int T, int D, int heads, float p_drop, unsigned long long seed_value, const unsigned long long* seed_base_dev,
                     void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!qkv_dev || !lens_dev || !out_dev || !dout_dev || !lse_dev || !dsum_dev || !dqkv_dev) return fail(EVMI_ERR_INVALID_ARG, "mha_bwd: null pointer");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads || p_drop < 0.f || p_drop >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "mha_bwd: shape / dropout");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "mha_bwd: grid limits");
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * heads * T;
  hipLaunchKernelGGL(attention_rowdot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out_dev, dout_dev, dsum_dev, B, T, heads, dh);
  const dim3 grid((T + 127) / 128, heads, B);
#define EVMI_MHA_BWD(DH, SLOT)                                                                                                    \
  {                                                                                                                               \
    constexpr size_t lds_q = (size_t)4 * DH * 32 * sizeof(float), lds_kv = lds_q + 128 * sizeof(float);                           \
    static thread_local bool configured_dev[kMaxDevices][3] = {};                                                \
    bool* configured = configured_dev[device_slot()];          \
    if (!configured[SLOT]) {                                                                                                      \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)attention_train_dq_kernel<DH>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                         (int)lds_q));                                                                            \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)attention_train_dkv_kernel<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds_kv));                                                                           \
      configured[SLOT] = true;                                                                                                    \
    }                                                                                                                             \
    hipLaunchKernelGGL(attention_train_dq_kernel<DH>, grid, dim3(256), lds_q, s, qkv_dev, lens_dev, dout_dev, lse_dev, dsum_dev,  \
                       dqkv_dev, B, T, D, scale, p_drop, seed);                                                                   \
    hipLaunchKernelGGL(attention_train_dkv_kernel<DH>, grid, dim3(256), lds_kv, s, qkv_dev, lens_dev, dout_dev, lse_dev, dsum_dev,\
                       dqkv_dev, B, T, D, scale, p_drop, seed);                                                                   \
  }
  if (dh == 128) EVMI_MHA_BWD(128, 0)
  else if (dh == 64) EVMI_MHA_BWD(64, 1)
  else if (dh == 32) EVMI_MHA_BWD(32, 2)
  else return fail(EVMI_ERR_UNSUPPORTED, "mha_bwd: head dimension must be 32, 64 or 128");
#undef EVMI_MHA_BWD
  EVMI_LAUNCH_CHECK("mha_bwd");
  return EVMI_OK;
}

}  // extern "C"
