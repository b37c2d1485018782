// Contrastive (InfoNCE) and ranking (BPR) losses, forward and backward, plus the gather / scatter-add of the
// minibatch rows.  Replaces Utils/Utils.py:57-98 and the index_select / index_add autograd pairs of
// Main.py:305-307 and Utils.py:67-68.
//
// InfoNCE never materialises the B x B logit matrix: one wave owns one anchor row, keeps its own normalised
// row in LDS, streams the other view's rows out of L2 and folds them into an online log-sum-exp (forward) or
// into a probability row held in LDS (backward).  [v1: fp32 VALU dot products; the MFMA tiling is the next step.]
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "common.h"

namespace dmm {

struct SlotTable {
  const float* src[16];
  const int32_t* idx[16];
  float* grad[16];
};
struct TermTable {
  dmm_infonce_term t[16];
};

// ---------------------------------------------------------------------------------------------
// zhat[s][b] = normalize(src_s[idx_s[b]]),  norm[s][b] = ||src_s[idx_s[b]]||
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bf16 operand packs for the MFMA InfoNCE kernels (Bp = B rounded up to 32, pad rows zero, lo = bf16(x - hi)).  Both are stored
// in MFMA-fragment order so that every operand load of a wave is ONE contiguous 512-byte access (32 lanes x 16 bytes per half):
//   hi/lo  [slot][row tile = b/32][ks = c/16][h][r = b%32][8]   element e <-> column 16 ks + 8 h + e      (A/B operand, k = column)
//   hiT/loT[slot][row tile = b/32][s][h][c][8]                  element e <-> row 32 tile + 16 s + 8 (e>>2) + 4 h + (e&3)
//                                                               (B operand of the X^T B product, k = batch row in accumulator order)
struct Bf16Pack {
  __bf16 *hi, *lo, *hiT, *loT;
  int Bp;
};

// LO = false (dmm_gather_normalize_hi: the caller's InfoNCE launches run at precision 1): only the hi / hiT blocks of the pack are written -- the lo / loT
// blocks, 5 of the 11 stores per lane, feed nothing but the split-precision kernels.
template <int D, bool LO = true>
__global__ __launch_bounds__(256) void gather_normalize_kernel(SlotTable S, int B, float* __restrict__ zhat, float* __restrict__ norm,
                                                                Bf16Pack pk) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int s = blockIdx.y;
  const int l = threadIdx.x % LANES;
  const float* __restrict__ src = S.src[s];
  const int32_t* __restrict__ idx = S.idx[s];
  const int rows = pk.hi ? pk.Bp : B;
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < rows; b += gridDim.x * G) {
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < B) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)idx[b] * D + 4 * l);
      const float nrm = sqrtf(group_sum<LANES>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
      const float inv = 1.0f / fmaxf(nrm, 1e-12f);
      y = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
      *reinterpret_cast<float4*>(zhat + ((size_t)s * B + b) * D + 4 * l) = y;
      if (l == 0) norm[(size_t)s * B + b] = nrm;
    }
    if (pk.hi) {
      const float f[4] = {y.x, y.y, y.z, y.w};
      __bf16 h[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)f[j];
        lo[j] = (__bf16)(f[j] - (float)h[j]);
      }
      const int tile = b >> 5, r = b & 31, c0 = 4 * l;
      // row-fragment order: columns c0..c0+3 share (ks, h) and are 4 consecutive elements
      const size_t o = ((((size_t)s * (pk.Bp >> 5) + tile) * (D / 16) + (c0 >> 4)) * 2 + ((c0 >> 3) & 1)) * 32 * 8 + (size_t)r * 8 + (c0 & 7);
      *reinterpret_cast<bf16x4*>(pk.hi + o) = bf16x4{h[0], h[1], h[2], h[3]};
      if (LO) *reinterpret_cast<bf16x4*>(pk.lo + o) = bf16x4{lo[0], lo[1], lo[2], lo[3]};
      // k-packed transposed order
      const int r4 = r & 15, ss = r >> 4, hh = (r4 >> 2) & 1, e = ((r4 >> 3) << 2) | (r4 & 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t ot = (((((size_t)s * (pk.Bp >> 5) + tile) * 2 + ss) * 2 + hh) * D + c0 + j) * 8 + e;
        pk.hiT[ot] = h[j];
        if (LO) pk.loT[ot] = lo[j];
      }
    }
  }
}

// grad_s[idx_s[b]] += normalize-backward(dzhat[s][b])      (fp32 atomics: duplicates in idx are legal)
template <int D>
__global__ __launch_bounds__(256) void scatter_normalize_bwd_kernel(SlotTable S, int B, const float* __restrict__ zhat,
                                                                     const float* __restrict__ norm, const float* __restrict__ dzhat) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int s = blockIdx.y;
  float* __restrict__ grad = S.grad[s];
  if (!grad) return;
  const int32_t* __restrict__ idx = S.idx[s];
  const int l = threadIdx.x % LANES;
  // a lane owns columns l, l + LANES, l + 2 LANES, l + 3 LANES: every atomic instruction then covers one contiguous 4 LANES-byte segment per
  // row (with 4 consecutive columns per lane it touched LANES scattered dwords of a 16 LANES-byte span: 60 -> ~25 us at B = 4096, 12 slots)
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < B; b += gridDim.x * G) {
    const size_t off = ((size_t)s * B + b) * D + l;
    float g[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { g[j] = dzhat[off + j * LANES]; y[j] = zhat[off + j * LANES]; }
    const float nrm = norm[(size_t)s * B + b];
    float o[4];
    if (nrm < 1e-12f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = g[j] * 1e12f;
    } else {
      const float dot = group_sum<LANES>(g[0] * y[0] + g[1] * y[1] + g[2] * y[2] + g[3] * y[3]);
      const float inv = 1.0f / nrm;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (g[j] - y[j] * dot) * inv;
    }
    float* dst = grad + (size_t)idx[b] * D + l;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(dst + j * LANES, o[j]);
  }
}

// ---------------------------------------------------------------------------------------------
// User-sharded batch exchange: a replica owns the user rows of the batch positions [seg[0], seg[1]) only.
//   gather_rows:       out[s][b] = src_s[idx_s[b]]            for b in the segment (the rest of `out` stays as the caller left it: zero,
//                                                             so that a sum all-reduce over the replicas assembles the batch table)
//   scatter_add_rows:  grad_s[idx_s[b]] += in[s][b]           for b in the segment (fp32 atomics: a user may repeat inside a batch)
// seg is read on the device (the launch plan stays static while the segment moves from step to step); seg == NULL: all of [0, B).
// ---------------------------------------------------------------------------------------------
template <int D, bool SCATTER>
__global__ __launch_bounds__(256) void rows_move_kernel(SlotTable S, int B, const int32_t* __restrict__ seg, float* __restrict__ buf) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int s = blockIdx.y, l = threadIdx.x % LANES;
  if (SCATTER && !S.grad[s]) return;  // a slot without a gradient table is skipped (its rows of `buf` are scattered by another launch)
  const int b0 = seg ? max(0, seg[0]) : 0, b1 = seg ? min(B, seg[1]) : B;
  const int32_t* __restrict__ idx = S.idx[s];
  for (int b = b0 + blockIdx.x * G + threadIdx.x / LANES; b < b1; b += gridDim.x * G) {
    float* t = buf + ((size_t)s * B + b) * D + 4 * l;
    if (idx[b] < 0) continue;  // a masked entry (item-sharded replicas: a row another replica owns): its table row is left as it is
    if (SCATTER) {  // columns l + j LANES per lane: contiguous atomic segments (scatter_normalize_bwd_kernel)
      float* dst = S.grad[s] + (size_t)idx[b] * D + l;
      const float* ts = buf + ((size_t)s * B + b) * D + l;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dst + j * LANES, ts[j * LANES]);
    } else {
      *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(S.src[s] + (size_t)idx[b] * D + 4 * l);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// InfoNCE forward: one wave per anchor row.
// ---------------------------------------------------------------------------------------------
__global__ void infonce_fwd_kernel(const float* __restrict__ zhat, TermTable T, int B, int D, float* __restrict__ lse,
                                   float* __restrict__ row_loss) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int waves = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.y;
  const int i_raw = blockIdx.x * waves + w;
  const bool valid = i_raw < B;
  const int i = valid ? i_raw : B - 1;  // surplus waves redo the last row and drop the result (keeps barriers uniform)
  const dmm_infonce_term term = T.t[t];
  const float* __restrict__ za = zhat + ((size_t)term.a * B + i) * D;
  const float* __restrict__ zb = zhat + (size_t)term.b * B * D;
  float* own = smem + (size_t)w * D;
  for (int c = lane; c < D; c += 64) own[c] = za[c];
  __syncthreads();
  const float inv_t = 1.0f / term.temp;
  float m = -INFINITY, lsum = 0.f, diag = 0.f;
  for (int j = lane; j < B; j += 64) {
    const float4* __restrict__ r = reinterpret_cast<const float4*>(zb + (size_t)j * D);
    float s = 0.f;
    for (int c = 0; c < D / 4; ++c) {
      const float4 x = r[c];
      const float4 o = reinterpret_cast<const float4*>(own)[c];
      s += x.x * o.x + x.y * o.y + x.z * o.z + x.w * o.w;
    }
    s *= inv_t;
    if (j == i) diag = s;
    const float mn = fmaxf(m, s);
    lsum = lsum * __expf(m - mn) + __expf(s - mn);
    m = mn;
  }
  const float M = wave_max(m);
  const float L = wave_sum(m == -INFINITY ? 0.f : lsum * __expf(m - M));
  diag = wave_sum(diag);
  if (lane == 0 && valid) {
    const float l = M + __logf(L);
    lse[(size_t)t * B + i] = l;
    row_loss[(size_t)t * B + i] = (l - diag) * (term.weight / (float)B);
  }
}

// InfoNCE backward.  blockIdx.z = 0: gradient of the anchor view (rows of the logit matrix);
//                    blockIdx.z = 1: gradient of the other view (columns).
__global__ void infonce_bwd_kernel(const float* __restrict__ zhat, const float* __restrict__ lse, TermTable T, int B, int D,
                                   float* __restrict__ dzhat) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int waves = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.y, pass = blockIdx.z;
  const int r_raw = blockIdx.x * waves + w;
  const bool valid = r_raw < B;
  const int r = valid ? r_raw : B - 1;
  const dmm_infonce_term term = T.t[t];
  const int own_slot = pass == 0 ? term.a : term.b, oth_slot = pass == 0 ? term.b : term.a;
  const float* __restrict__ zo = zhat + ((size_t)own_slot * B + r) * D;
  const float* __restrict__ zx = zhat + (size_t)oth_slot * B * D;
  const float* __restrict__ ls = lse + (size_t)t * B;
  float* own = smem + (size_t)w * ((D + B + 3) & ~3);  // 16-byte aligned per-wave slab
  float* prob = own + D;
  for (int c = lane; c < D; c += 64) own[c] = zo[c];
  __syncthreads();
  const float inv_t = 1.0f / term.temp;
  const float lse_r = ls[r];
  for (int k = lane; k < B; k += 64) {
    const float4* __restrict__ row = reinterpret_cast<const float4*>(zx + (size_t)k * D);
    float s = 0.f;
    for (int c = 0; c < D / 4; ++c) {
      const float4 x = row[c];
      const float4 o = reinterpret_cast<const float4*>(own)[c];
      s += x.x * o.x + x.y * o.y + x.z * o.z + x.w * o.w;
    }
    prob[k] = __expf(s * inv_t - (pass == 0 ? lse_r : ls[k])) - (k == r ? 1.f : 0.f);
  }
  __syncthreads();
  if (!valid) return;
  const float coef = term.weight * inv_t / (float)B;
  float* dst = dzhat + ((size_t)own_slot * B + r) * D;
  for (int c = lane; c < D; c += 64) {
    float acc = 0.f;
    for (int k = 0; k < B; ++k) acc = fmaf(prob[k], zx[(size_t)k * D + c], acc);
    atomicAdd(dst + c, acc * coef);
  }
}

// ---------------------------------------------------------------------------------------------
// InfoNCE on the matrix cores (D % 32 == 0).  One wave owns 32 "own" rows and walks the other view in tiles of 32.
// The logit tile is computed TRANSPOSED (rows = other index in the 16 accumulator registers, column = own index on
// the lane), so the softmax statistics of an own row are lane-local, and the probability tile, converted to bf16 in
// place, is directly the A operand (X^T B form) of the MFMA that accumulates d(own) = sum_k P[k][own] * other[k][:] --
// no LDS, no cross-lane traffic.  PREC = 3 adds the hi*lo + lo*hi products (operands split as hi + lo bf16).
// ---------------------------------------------------------------------------------------------
// XCD-aware unit mapping of the MFMA kernels below.  A "unit" is one walk target -- a term (forward) or a (term, pass) of the backward -- whose
// workgroups all stream the same "other" view.  Workgroups go to the 8 XCDs round-robin by linear id (MI355X_MICROARCH.md), each XCD with its own
// 4 MB L2; with the natural (row block, term, pass) grid every XCD had all terms in flight at once -- 8 views of 1 MB each against 4 MB of L2, and the
// walk ran at the Infinity Cache's rate (round 4: 512 MB of LDS-DMA per backward launch in 95-115 us).  Here unit u lives on XCD u % 8: the
// workgroups resident on an XCD walk ONE view (two while a unit drains), which stays in that L2.
struct UnitMap { int unit, sub; };
__device__ __forceinline__ UnitMap xcd_unit(int per_unit) {
  const int id = blockIdx.x, q = id >> 3;
  return UnitMap{(q / per_unit) * 8 + (id & 7), q % per_unit};
}
static inline unsigned xcd_grid(int n_units, int per_unit) { return 8u * (unsigned)((n_units + 7) / 8) * (unsigned)per_unit; }

// fragment (ks, h) of row r of a 32-row tile in the row-fragment order
template <int D>
__device__ __forceinline__ size_t frag_off(int slot, int tiles, int tile, int ks, int h, int r) {
  return ((((size_t)slot * tiles + tile) * (D / 16) + ks) * 2 + h) * 256 + (size_t)r * 8;
}

template <int D, int PREC>
__device__ __forceinline__ f32x16 logits_tile(const __bf16* __restrict__ hi, const __bf16* __restrict__ lo, int slot, int tiles, int tile,
                                              const bf16x8 (&own_h)[D / 16], const bf16x8 (&own_l)[D / 16], int fr, int fh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(slot, tiles, tile, ks, fh, fr);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(hi + o);
    if (PREC == 3) {
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(lo + o);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, own_h[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, own_l[ks], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, own_h[ks], acc, 0, 0, 0);
  }
  return acc;
}

template <int D, int PREC>
__global__ __launch_bounds__(256) void infonce_fwd_mfma_kernel(Bf16Pack pk, TermTable T, int n_terms, int B, float* __restrict__ lse,
                                                                float* __restrict__ row_loss) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
  // One workgroup = 32 own rows; its 4 waves walk different quarters of the other view and merge their (max, sum, diagonal) through
  // LDS -- 4x the waves of a one-wave-per-row-block layout (which left one wave per SIMD, nothing to overlap its load latency with).
  __shared__ float part[4][32][3];
  const UnitMap um = xcd_unit((B + 31) >> 5);  // unit = term, sub = block of 32 own rows
  if (um.unit >= n_terms) return;
  const int t = um.unit;
  const int i0 = um.sub * 32;
  const dmm_infonce_term term = T.t[t];
  const int Bp = pk.Bp;
  const int tiles = Bp >> 5;
  bf16x8 own_h[D / 16], own_l[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(term.a, tiles, i0 >> 5, ks, fh, fr);
    own_h[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
    own_l[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
  }
  const float inv_t = 1.0f / term.temp;
  const float c2 = inv_t * 1.4426950408889634f;  // logits enter exp2 as s * c2
  const int i = i0 + fr;
  float m = -INFINITY, ssum = 0.f, diag = 0.f;  // m: running maximum of s * c2 (base-2 domain)
  bf16x8 ca[D / 16], cl[D / 16], na[D / 16], nl[D / 16];
  const int n_t = (B + 31) >> 5, per = (n_t + 3) >> 2;
  const int jt0 = wave * per, jt1 = min(n_t, jt0 + per);
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(term.b, tiles, min(jt0, n_t - 1), ks, fh, fr);
    ca[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
    if (PREC == 3) cl[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
  }
  for (int j0 = jt0 * 32; j0 < jt1 * 32; j0 += 32) {
    if (j0 + 32 < jt1 * 32) {  // next tile's operand: in flight under this tile's math
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const size_t o = frag_off<D>(term.b, tiles, (j0 >> 5) + 1, ks, fh, fr);
        na[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
        if (PREC == 3) nl[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      if (PREC == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[ks], own_h[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[ks], own_l[ks], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[ks], own_h[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      ca[ks] = na[ks];
      if (PREC == 3) cl[ks] = nl[ks];
    }
    // online log-sum-exp in the base-2 domain: m, mn are maxima of s * c (c = log2(e) / temp), p = exp2(fma(s, c, -mn)) -- one fmax, one
    // fma, one v_exp and one add per logit.  The bounds test runs only on the last tile, the diagonal pick only on the tile that holds it
    // (13 -> 7 issue slots per logit: the kernel is VALU-bound).
    if (j0 == i0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + 4 * fh == i) diag = acc[r] * inv_t;
    }
    if (j0 + 32 > B) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + 4 * fh >= B) acc[r] = -INFINITY;
    }
    float tmax = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, acc[r]);
    if (tmax > -INFINITY) {
      const float mn = fmaxf(m, tmax * c2);
      float add = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mn));
      ssum = ssum * __builtin_amdgcn_exp2f(m - mn) + add;
      m = mn;
    }
  }
  // the two lane halves hold disjoint column subsets of the same own row
  const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64), d2 = __shfl_xor(diag, 32, 64);
  const float M = fmaxf(m, m2);
  const float L = (m > -INFINITY ? ssum * __builtin_amdgcn_exp2f(m - M) : 0.f) + (m2 > -INFINITY ? s2 * __builtin_amdgcn_exp2f(m2 - M) : 0.f);
  if (fh == 0) {
    part[wave][fr][0] = M;
    part[wave][fr][1] = L;
    part[wave][fr][2] = diag + d2;
  }
  __syncthreads();
  if (wave == 0 && fh == 0 && i < B) {
    float Mx = -INFINITY, dg = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      Mx = fmaxf(Mx, part[w][fr][0]);
      dg += part[w][fr][2];
    }
    float Ls = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (part[w][fr][0] > -INFINITY) Ls += part[w][fr][1] * __builtin_amdgcn_exp2f(part[w][fr][0] - Mx);
    const float l = (Mx + __log2f(Ls)) * 0.6931471805599453f;  // back to natural units
    lse[(size_t)t * B + i] = l;
    row_loss[(size_t)t * B + i] = (l - dg) * (term.weight / (float)B);
  }
}

// blockIdx.z = 2 * slice + pass.  pass 0: own = anchor view a (lse indexed by the own row); pass 1: own = view b (lse indexed by
// the other row).  The walk over the other view is cut into KSPLIT slices handled by different waves (the probabilities only
// need the precomputed lse, and the output is accumulated with atomics anyway): 4x more waves in flight hides the L2 latency of
// the per-tile operand loads, which are also issued one tile ahead.
constexpr int INFONCE_KSPLIT = 4;

template <int D, int PREC>
struct OthFrags {
  bf16x8 a_h[D / 16], a_l[D / 16];        // logit-tile A operand (row-fragment order)
  bf16x8 t_h[2][D / 32], t_l[2][D / 32];  // output-product B operand (k-packed transposed order)
};

template <int D, int PREC>
__device__ __forceinline__ void load_oth(OthFrags<D, PREC>& f, const Bf16Pack& pk, int slot, int tiles, int tile, int fr, int fh) {
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(slot, tiles, tile, ks, fh, fr);
    f.a_h[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
    if (PREC == 3) f.a_l[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb) {
      const size_t ot = (((((size_t)slot * tiles + tile) * 2 + s) * 2 + fh) * D + cb * 32 + fr) * 8;
      f.t_h[s][cb] = *reinterpret_cast<const bf16x8*>(pk.hiT + ot);
      if (PREC == 3) f.t_l[s][cb] = *reinterpret_cast<const bf16x8*>(pk.loT + ot);
    }
}

template <int D, int PREC>
__global__ __launch_bounds__(256) void infonce_bwd_mfma_kernel(Bf16Pack pk, const float* __restrict__ lse, TermTable T, int B,
                                                                float* __restrict__ dzhat) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
  const int t = blockIdx.y, pass = blockIdx.z & 1, slice = blockIdx.z >> 1;
  const int o0 = (blockIdx.x * 4 + wave) * 32;
  if (o0 >= B) return;
  const dmm_infonce_term term = T.t[t];
  const int own_slot = pass == 0 ? term.a : term.b, oth_slot = pass == 0 ? term.b : term.a;
  const int tiles = pk.Bp >> 5;
  const int ksplit = gridDim.z >> 1;
  const int per = (tiles + ksplit - 1) / ksplit;
  const int kt0 = slice * per, kt1 = min(tiles, kt0 + per);
  if (kt0 >= kt1) return;
  bf16x8 own_h[D / 16], own_l[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(own_slot, tiles, o0 >> 5, ks, fh, fr);
    own_h[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
    own_l[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
  }
  f32x16 out[D / 32];
#pragma unroll
  for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[cb][r] = 0.f;
  const float inv_t = 1.0f / term.temp;
  const float* __restrict__ ls = lse + (size_t)t * B;
  const int oidx = o0 + fr;
  const float lse_own = ls[oidx < B ? oidx : B - 1];
  OthFrags<D, PREC> cur, nxt;
  load_oth<D, PREC>(cur, pk, oth_slot, tiles, kt0, fr, fh);
  for (int kt = kt0; kt < kt1; ++kt) {
    if (kt + 1 < kt1) load_oth<D, PREC>(nxt, pk, oth_slot, tiles, kt + 1, fr, fh);  // in flight under this tile's math
    const int k0 = kt * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      if (PREC == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a_l[ks], own_h[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a_h[ks], own_l[ks], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a_h[ks], own_h[ks], acc, 0, 0, 0);
    }
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kidx = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const float l = pass == 0 ? lse_own : ls[kidx < B ? kidx : B - 1];
      p[r] = kidx < B ? __expf(acc[r] * inv_t - l) - (kidx == oidx ? 1.f : 0.f) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // k-step s of the output MFMA covers tile rows 16s .. 16s+15 in accumulator order
      bf16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ph[j] = (__bf16)p[8 * s + j];
        if (PREC == 3) pl[j] = (__bf16)(p[8 * s + j] - (float)ph[j]);
      }
#pragma unroll
      for (int cb = 0; cb < D / 32; ++cb) {
        if (PREC == 3) {
          out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, cur.t_h[s][cb], out[cb], 0, 0, 0);
          out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, cur.t_l[s][cb], out[cb], 0, 0, 0);
        }
        out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, cur.t_h[s][cb], out[cb], 0, 0, 0);
      }
    }
    if (kt + 1 < kt1) cur = nxt;
  }
  const float coef = term.weight * inv_t / (float)B;
#pragma unroll
  for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = o0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (row < B) atomicAdd(dzhat + ((size_t)own_slot * B + row) * D + cb * 32 + fr, out[cb][r] * coef);
    }
}

// The same backward with the walk over the other view SHARED by the four waves of a workgroup: every 32-row tile of the other view
// (its row-fragment block for the logits and its k-packed transposed block for the output product: 2 x 32 x D bf16, contiguous in the
// pack) is brought into an LDS ring by LDS-DMA (global_load_lds, no staging registers), NS - 1 tiles ahead of the math, and read by
// all four waves -- a quarter of the L2 traffic of the per-wave walk above, which was latency-bound on exactly those loads (MFMA pipe
// 12 % busy).  Completion is counted (s_waitcnt vmcnt((NS - 2) * PER): the younger tiles stay in flight across the raw s_barrier).
// Each wave owns its 32 rows over the WHOLE walk, so the K split and its 4x atomics are gone: one atomicAdd per output element and
// (term, pass) -- a slot still takes part in several terms.
template <int D, int PREC>
struct InbShape {
  static constexpr int NB = PREC == 3 ? 4 : 2;   // blocks per tile: hi, hiT (+ lo, loT)
  static constexpr int BLK = 32 * D;             // bf16 elements per block
  static constexpr int TILE = NB * BLK;
  static constexpr int PER = TILE / 8 / 256;     // 16-byte LDS-DMA pieces per lane and tile
  static constexpr int NS = TILE * 2 * 4 <= 131072 ? 4 : (TILE * 2 * 3 <= 131072 ? 3 : 2);  // ring stages
  static constexpr int LDS_BYTES = NS * TILE * 2;
};
template <int N>
__device__ __forceinline__ void inb_wait_vmcnt() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // conservative for shapes without a dedicated immediate
}

// One 32-row tile of the other view (LDS image S) against the wave's 32 own rows, in two halves so that the walk can run them one tile apart
// (the logits of tile k + 1 are on the matrix pipe while the VALU turns tile k's into probabilities):
//  * tile_logits: logits, transposed (other index in the accumulator registers, own index on the lane);
//  * inb_apply:   probabilities minus the identity and their product with the other rows, added into `out`.
// CHECK = false drops the per-element bounds and diagonal tests (two compares and two selects per element of a VALU-bound loop).
// The LDS fragments of a tile as values, so that a step can read ALL of them right behind its barrier -- one LDS round trip per step instead of
// five (round-4 ablation: with the exponentials AND every MFMA removed the backward still took 82 of its 92 us; without the LDS reads 23 us: the
// walk was bound by the latency of its serial LDS round trips, not by the matrix or the transcendental rate).
template <int D, int PREC>
struct SFrags { bf16x8 h[D / 16], l[PREC == 3 ? D / 16 : 1]; };        // row-fragment blocks: the logits' A operand
template <int D, int PREC>
struct TFrags { bf16x8 h[2][D / 32], l[PREC == 3 ? 2 : 1][D / 32]; };  // k-packed transposed blocks: the product's B operand
template <int D, int PREC, int LO = 2>  // LO: index of the `lo` block within the tile image
__device__ __forceinline__ SFrags<D, PREC> load_sfrags(const __bf16* S, int fr, int fh) {
  constexpr int BLK = 32 * D;
  SFrags<D, PREC> f;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const int fo = ((ks * 2 + fh) * 32 + fr) * 8;
    f.h[ks] = *reinterpret_cast<const bf16x8*>(S + fo);
    if (PREC == 3) f.l[ks] = *reinterpret_cast<const bf16x8*>(S + LO * BLK + fo);
  }
  return f;
}
template <int D, int PREC>
__device__ __forceinline__ TFrags<D, PREC> load_tfrags(const __bf16* S, int fr, int fh) {
  constexpr int BLK = 32 * D;
  TFrags<D, PREC> f;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb) {
      const int to = ((s2 * 2 + fh) * D + cb * 32 + fr) * 8;
      f.h[s2][cb] = *reinterpret_cast<const bf16x8*>(S + BLK + to);
      if (PREC == 3) f.l[s2][cb] = *reinterpret_cast<const bf16x8*>(S + 3 * BLK + to);
    }
  return f;
}
template <int D, int PREC>
__device__ __forceinline__ f32x16 logits_from(const SFrags<D, PREC>& f, const bf16x8 (&own_h)[D / 16], const bf16x8 (&own_l)[D / 16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    if (PREC == 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l[ks], own_h[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h[ks], own_l[ks], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h[ks], own_h[ks], acc, 0, 0, 0);
  }
  return acc;
}
template <int D, int PREC, int LO = 2>
__device__ __forceinline__ f32x16 tile_logits(const __bf16* S, const bf16x8 (&own_h)[D / 16], const bf16x8 (&own_l)[D / 16], int fr, int fh) {
  return logits_from<D, PREC>(load_sfrags<D, PREC, LO>(S, fr, fh), own_h, own_l);
}
// P (16 probabilities per lane, accumulator order) times the tile's other rows, added into `out`
template <int D, int PREC>
__device__ __forceinline__ void product_from(const TFrags<D, PREC>& f, const float (&p)[16], f32x16 (&out)[D / 32]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {  // k-step s2 of the output MFMA covers tile rows 16 s2 .. 16 s2 + 15 in accumulator order
    bf16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ph[j] = (__bf16)p[8 * s2 + j];
      if (PREC == 3) pl[j] = (__bf16)(p[8 * s2 + j] - (float)ph[j]);
    }
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb) {
      if (PREC == 3) {
        out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, f.h[s2][cb], out[cb], 0, 0, 0);
        out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, f.l[s2][cb], out[cb], 0, 0, 0);
      }
      out[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, f.h[s2][cb], out[cb], 0, 0, 0);
    }
  }
}

// The other rows' log-sum-exp values (pass 1, interior tiles) are read by INLINE ASSEMBLY: as C++ loads from the dynamic LDS array that also holds
// the LDS-DMA ring, the compiler cannot rule out that they alias an LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of them -- every tile of
// the walk then drained the prefetch (found in the round-4 disassembly).  lse_issue starts the reads, lse_wait is their explicit wait.
// `ls_lds` is indexed by the other row (the caller passes the split's vector shifted so that the view's row index applies).
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct LseRegs { f32x4v v0, v1, v2, v3; };
__device__ __forceinline__ LseRegs lse_issue(const float* ls_lds, int k0, int fh) {
  typedef __attribute__((address_space(3))) const void* lds_cp;
  LseRegs r;
  const unsigned la = (unsigned)(uintptr_t)(lds_cp)(ls_lds + k0 + 4 * fh);
  asm volatile("ds_read_b128 %0, %1" : "=v"(r.v0) : "v"(la));
  asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(r.v1) : "v"(la));
  asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(r.v2) : "v"(la));
  asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(r.v3) : "v"(la));
  return r;
}
__device__ __forceinline__ void lse_wait(LseRegs& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v0), "+v"(r.v1), "+v"(r.v2), "+v"(r.v3)::"memory"); }

// probabilities minus the identity, times the other rows (fragments `tf`), added into `out`; `lr`: lse_issue's registers (plain pass-1 tiles only)
template <int D, int PREC, bool CHECK, bool own_lse>
__device__ __forceinline__ void inb_apply(const TFrags<D, PREC>& tf, const f32x16& acc, f32x16 (&out)[D / 32], int k0, int oidx, int B, float c1,
                                          float l2_own, const float* ls_lds, const LseRegs& lr, int fh) {
  float p[16], l2[16];
  if (own_lse) {
#pragma unroll
    for (int r = 0; r < 16; ++r) l2[r] = l2_own;
  } else if (!CHECK) {  // accumulator registers 4g .. 4g+3 hold four consecutive other rows: one 16-byte LDS read each (waited for by the caller)
#pragma unroll
    for (int j = 0; j < 4; ++j) { l2[j] = lr.v0[j]; l2[4 + j] = lr.v1[j]; l2[8 + j] = lr.v2[j]; l2[12 + j] = lr.v3[j]; }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kidx = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      l2[r] = ls_lds[kidx < B ? kidx : B - 1];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kidx = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    p[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c1, -l2[r]));
    if (CHECK) p[r] = kidx < B ? p[r] - (kidx == oidx ? 1.f : 0.f) : 0.f;
  }
  product_from<D, PREC>(tf, p, out);
}

// R = own 32-row tiles per wave.  R = 2 (d <= 64, one bf16 pass): every LDS fragment, lse value, LDS-DMA piece and barrier of a step serves 2048
// logits per wave instead of 1024 -- the walk is bound by its per-step latencies (barrier across four SIMDs, LDS round trip, the dependent MFMA
// chain: round-4 stamps and ablations in DESIGN 4.3), not by the matrix, transcendental or LDS rates, so the lever is more work per step.
template <int D, int PREC>
struct InbRows { static constexpr int R = (D <= 64 && PREC == 1) ? 2 : 1; };

template <int D, int PREC>
__global__ __launch_bounds__(256) void infonce_bwd_lds_kernel(Bf16Pack pk, const float* __restrict__ lse, TermTable T, int n_terms, int n_pass, int B,
                                                               float* __restrict__ dzhat, int pass_base, int ksplit) {
  using Sh = InbShape<D, PREC>;
  constexpr int BLK = Sh::BLK, TILE = Sh::TILE, PER = Sh::PER, NS = Sh::NS, R = InbRows<D, PREC>::R;
  static_assert(TILE % (8 * 256) == 0, "tile must split into whole 16-byte pieces per lane");
  extern __shared__ __attribute__((aligned(16))) __bf16 sm_oth[];  // [NS][TILE]
  typedef __attribute__((address_space(3))) void* lds_p;
  typedef const __attribute__((address_space(1))) void* gbl_p;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  // (wave index through readfirstlane: the walk's counters, the own tile's index and the LDS-DMA source block are then scalar -- as a vector
  // value the compiler ran the walk under exec-mask control flow)
  // unit = pass * n_terms + term (one per XCD at a time, see xcd_unit), sub = split * row blocks + row block; a split walks its share of the other
  // view's tiles (the probabilities are normalised by lse, so the splits' partial products simply add: one more atomicAdd per output element and split)
  const int nblk = (B + 128 * R - 1) / (128 * R);
  const UnitMap um = xcd_unit(nblk * ksplit);
  if (um.unit >= n_terms * n_pass) return;  // (whole workgroups, before the first barrier)
  const int t = um.unit % n_terms, pass = um.unit / n_terms + pass_base, split = um.sub / nblk;
  const int o0 = ((um.sub % nblk) * 4 + wave) * 32 * R;
  const bool active = o0 < B;
  const dmm_infonce_term term = T.t[t];
  const int own_slot = pass == 0 ? term.a : term.b, oth_slot = pass == 0 ? term.b : term.a;
  const int tiles = pk.Bp >> 5;
  const int kt_beg = (int)((long long)tiles * split / ksplit), kt_end = (int)((long long)tiles * (split + 1) / ksplit);
  const size_t oth_base = (size_t)oth_slot * tiles * BLK;

  // piece i of a tile: 16 bytes per lane, pieces of one wave-instruction contiguous (what LDS-DMA writes: base + 16 * lane); a wave's 64 chunks lie in
  // one block (64 | BLK / 8), so the source of piece i is a wave-uniform base (chosen once, outside the walk) + k * BLK + 8 * lane
  const __bf16* srcb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c0 = i * 256 + wave * 64, blk = c0 / (BLK / 8);
    srcb[i] = (blk == 0 ? pk.hi : blk == 1 ? pk.hiT : blk == 2 ? pk.lo : pk.loT) + oth_base + (size_t)(c0 % (BLK / 8)) * 8;
  }
  auto issue = [&](int kt, int stage) {
    const int k = kt < tiles ? kt : tiles - 1;  // past the end: a redundant copy of the last tile (keeps the DMA count per iteration fixed)
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds((gbl_p)(srcb[i] + (size_t)k * BLK + (size_t)lane * 8),
                                       (lds_p)(sm_oth + (size_t)stage * TILE + (size_t)(i * 256 + wave * 64) * 8), 16, 0, 0);
  };
  static_assert((BLK / 8) % 64 == 0, "a wave-instruction of the LDS-DMA must not straddle two blocks");

  // own tiles otile .. otile + R - 1 (a tile past the view: the last one again -- its rows are not written)
  bf16x8 own_h[R][D / 16], own_l[R][D / 16];
  const int otile = active ? (o0 >> 5) : 0;
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const size_t o = frag_off<D>(own_slot, tiles, otile + q < tiles ? otile + q : tiles - 1, ks, fh, fr);
      own_h[q][ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
      if (PREC == 3) own_l[q][ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
    }
  f32x16 out[R][D / 32];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[q][cb][r] = 0.f;
  const float inv_t = 1.0f / term.temp;
  const float* __restrict__ ls = lse + (size_t)t * B;
  constexpr float LOG2E = 1.4426950408889634f;
  const float c1 = inv_t * LOG2E;  // p = exp(s / temp - lse) = exp2(s * c1 - lse * log2 e): one fma + one v_exp
  int oidx[R];
  float l2_own[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    oidx[q] = o0 + 32 * q + fr;
    l2_own[q] = ls[oidx[q] < B ? oidx[q] : B - 1] * LOG2E;
  }
  // pass 1 indexes the log-sum-exp by the OTHER row: the rows this split walks go to LDS once, so that the walk issues no register loads (a
  // waited register load would drain the LDS-DMA queue: vector-memory operations retire in order); indexed by the view's row through a shifted base
  float* const ls_lds = reinterpret_cast<float*>(sm_oth + (size_t)NS * TILE) - (size_t)kt_beg * 32;
  if (pass == 1)
    for (int i = kt_beg * 32 + tid; i < kt_end * 32; i += 256) ls_lds[i] = ls[i < B ? i : B - 1] * LOG2E;
  // the register loads above are out of the counted window, and the LDS stores have landed before the first (raw) barrier
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#pragma unroll
  for (int j = 0; j < NS - 1; ++j) issue(kt_beg + j, j);
  // One step of the walk.  Every wave of the workgroup runs the same NUMBER of steps (the raw barrier counts them), each in the form its own
  // rows need.  Bounds / diagonal handling only where it can matter -- the tiles that hold the wave's own rows and the (padded) last tile -- and as
  // SEPARATE loops: with `if (special) tile<true> else tile<false>` inside one loop the compiler merged the two bodies (round-4 disassembly: the 64
  // compares / selects of the checked form and a 48-instruction AGPR <-> VGPR shuffle of `out` ran on every tile, next to 16 v_exp + 16 fma).
  // PIPE (ring of >= 3 stages): a step issues the logits MFMAs of tile kt + 1 and THEN turns tile kt's (formed one step earlier) into
  // probabilities; EARLY: every LDS read of the step is issued right behind the barrier (one LDS round trip per step).
  constexpr bool PIPE = NS >= 3;
  constexpr bool EARLY = D <= 64;
  auto walk = [&](auto own_lse) {  // own_lse: pass 0, the probabilities are normalised by the OWN row's log-sum-exp (a lane constant)
    int kt = kt_beg;
    f32x16 acc_a[R], acc_b[R];  // PIPE: logits of the current tile / of the next one (the roles alternate: no copy in the plain loop)
    if (PIPE) {
      inb_wait_vmcnt<(NS - 2) * PER>();
      asm volatile("s_barrier" ::: "memory");
      const SFrags<D, PREC> sf = load_sfrags<D, PREC>(sm_oth, fr, fh);
#pragma unroll
      for (int q = 0; q < R; ++q) acc_a[q] = logits_from<D, PREC>(sf, own_h[q], own_l[q]);
    }
    auto step = [&](auto checked, f32x16 (&cur)[R], f32x16 (&nxt)[R]) {
      constexpr bool CHK = decltype(checked)::value, OWN = decltype(own_lse)::value;
      const int rel = kt - kt_beg;
      // this wave's pieces of tile kt (PIPE: kt + 1) have landed; the younger tiles stay in flight
      inb_wait_vmcnt<(PIPE ? NS - 3 : NS - 2) * PER>();
      asm volatile("s_barrier" ::: "memory");  // ... every wave's pieces; and every wave is done reading tile kt - 1
      issue(kt + NS - 1, (rel + NS - 1) % NS);  // into the stage tile kt - 1 occupied
      const __bf16* S = sm_oth + (size_t)(rel % NS) * TILE;
      LseRegs lr{};
      if (PIPE) {
        // in the order of use: next tile's row fragments, the lse values, this tile's transposed fragments
        const SFrags<D, PREC> sf = load_sfrags<D, PREC>(sm_oth + (size_t)((rel + 1) % NS) * TILE, fr, fh);
        if (!OWN && !CHK) lr = lse_issue(ls_lds, kt * 32, fh);
        const TFrags<D, PREC> tf = load_tfrags<D, PREC>(S, fr, fh);
        if (EARLY) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < R; ++q) nxt[q] = logits_from<D, PREC>(sf, own_h[q], own_l[q]);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
          if (!OWN && !CHK) lse_wait(lr);
#pragma unroll
          for (int q = 0; q < R; ++q) inb_apply<D, PREC, CHK, OWN>(tf, cur[q], out[q], kt * 32, oidx[q], B, c1, l2_own[q], ls_lds, lr, fh);
        }
      } else if (active) {
        const SFrags<D, PREC> sf = load_sfrags<D, PREC>(S, fr, fh);
        if (!OWN && !CHK) { lr = lse_issue(ls_lds, kt * 32, fh); lse_wait(lr); }
        const TFrags<D, PREC> tf = load_tfrags<D, PREC>(S, fr, fh);
#pragma unroll
        for (int q = 0; q < R; ++q) {
          cur[q] = logits_from<D, PREC>(sf, own_h[q], own_l[q]);
          inb_apply<D, PREC, CHK, OWN>(tf, cur[q], out[q], kt * 32, oidx[q], B, c1, l2_own[q], ls_lds, lr, fh);
        }
      }
      ++kt;
    };
    auto copy = [&] {
#pragma unroll
      for (int q = 0; q < R; ++q) acc_a[q] = acc_b[q];
    };
#pragma unroll 1
    for (int phase = 0; phase <= R; ++phase) {
      // phases 0 .. R - 1: plain tiles up to own tile `phase`, then that tile checked (a split that starts behind it checks its first tile instead:
      // harmless); phase R: plain tiles up to the last tile of the view, then that one checked if this split holds it
      const int special = phase < R ? otile + phase : tiles - 1;
      const int upto = special < kt ? kt : (special > kt_end ? kt_end : special);
#pragma unroll 1
      while (kt + 1 < upto) {
        step(std::false_type{}, acc_a, acc_b);
        step(std::false_type{}, acc_b, acc_a);
      }
      if (kt < upto) {
        step(std::false_type{}, acc_a, acc_b);
        copy();
      }
      if (kt < kt_end) {
        step(std::true_type{}, acc_a, acc_b);
        copy();
      }
    }
  };
  if (pass == 0) walk(std::true_type{});
  else walk(std::false_type{});
  // (every wave has taken kt_end - kt_beg steps: the last phase ends at min(max(tiles - 1, kt), kt_end) and its checked step takes the last tile if it
  // is in range)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing redundant copies have landed before the workgroup's LDS goes away
  if (!active) return;
  const float coef = term.weight * inv_t / (float)B;
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = o0 + 32 * q + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < B) atomicAdd(dzhat + ((size_t)own_slot * B + row) * D + cb * 32 + fr, out[q][cb][r] * coef);
      }
}

constexpr int INB_MAX_B = 8192;  // the lse values of a split's rows (4 B per batch row) sit in LDS behind the ring: 32 KiB at most
template <int D, int PREC>
static void inb_launch(int n_terms, int n_pass, hipStream_t s, Bf16Pack pk, const float* lse, const TermTable& T, int B, float* dzhat, int pass_base = 0) {
  // walk splits per (term, pass).  Before the XCD-aware unit mapping: 1 / 2 / 4 / 8 measured 123.5 / 117.6 / 113.8 / ~125 us at B = 4096, 8 terms.
  // Now a unit's workgroups should fill the XCD that owns it once: row blocks x splits = what the XCD holds.
  static const int forced = [] { const char* e = getenv("DMM_INFONCE_LDS_KSPLIT"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
  constexpr int R = InbRows<D, PREC>::R;
  const int nblk = cdiv(B, 128 * R), tiles = (B + 31) >> 5;
  // workgroups an XCD holds at once: 32 CUs x min(registers: 4 / R, LDS: ring + ~a quarter of the lse vector)
  const int per_cu = std::max(1, std::min(4 / R, (int)((160 * 1024) / (InbShape<D, PREC>::LDS_BYTES + (size_t)cdiv(tiles, 4) * 128))));
  int ksplit = forced ? forced : (16 * per_cu + nblk / 2) / nblk;  // (B = 4096, d = 64: 1 / 2 / 3 / 4 / 6 splits -> 100 / 88 / 93 / 99 / 109 us)
  ksplit = ksplit < 1 ? 1 : (ksplit > 8 ? 8 : ksplit);
  if (ksplit > tiles) ksplit = tiles;
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(infonce_bwd_lds_kernel<D, PREC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              InbShape<D, PREC>::LDS_BYTES + INB_MAX_B * 4);
    return true;
  }();
  (void)once;
  const size_t lds_bytes = InbShape<D, PREC>::LDS_BYTES + (size_t)cdiv(tiles, ksplit) * 32 * 4;  // ring + the lse values of one split's rows
  hipLaunchKernelGGL((infonce_bwd_lds_kernel<D, PREC>), dim3(xcd_grid(n_terms * n_pass, nblk * ksplit)), dim3(256), lds_bytes, s, pk, lse, T, n_terms,
                     n_pass, B, dzhat, pass_base, ksplit);
}

// ---- forward + the anchor-side backward in one walk (unit-norm rows) -------------------------------------------------------------------
// With cosine-normalised views every logit s <= 1, so exp(s / temp) needs no running maximum: p' = exp2(s c1 - c1) <= 1 with
// c1 = log2(e) / temp.  One walk over the other view then yields, per own (anchor) row i: L_i = sum_k p'_ik (-> lse_i = ln2 (c1 + log2 L_i),
// the forward's output), the raw diagonal logit (-> row_loss_i), and N_i = sum_k p'_ik other_k (the MFMA product the backward's pass 0
// accumulates) -> d anchor_i = (N_i / L_i - other_i) w / (temp B): the forward kernel and HALF of the backward (its pass 0 recomputed every
// logit and exponential) become one kernel; the backward keeps only pass 1 (gradient w.r.t. the other view, which needs lse_i first).
// PRODUCT = false is the forward alone (lse, row loss): the same walk without the product, over tiles that hold only the row-fragment blocks.
template <int D, int PREC, bool PRODUCT>
struct WalkShape {
  static constexpr int NB = (PRODUCT ? 2 : 1) * (PREC == 3 ? 2 : 1);  // PRODUCT: hi, hiT (, lo, loT) as InbShape; else hi (, lo)
  static constexpr int BLK = 32 * D;
  static constexpr int LO = PRODUCT ? 2 : 1;      // block index of `lo`
  static constexpr int TILE = NB * BLK;
  static constexpr int PER = TILE / 8 / 256;
  static constexpr int NS = TILE * 2 * 4 <= 131072 ? 4 : (TILE * 2 * 3 <= 131072 ? 3 : 2);
  static constexpr int LDS_BYTES = NS * TILE * 2;
  static constexpr bool OK = TILE % (8 * 256) == 0 && NS >= 3;  // whole LDS-DMA pieces per lane, and a ring deep enough for the pipelined walk
};

template <int D, int PREC, bool CHECK, bool PRODUCT>
__device__ __forceinline__ void inf_apply(const TFrags<D, PREC>& tf, const f32x16& acc, f32x16 (&out)[D / 32], int k0, int oidx, int B, float c1, float& lsum,
                                          float& diag, int fh) {
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c1, -c1));
    if (CHECK) {
      const int kidx = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (kidx >= B) p[r] = 0.f;
      if (kidx == oidx) diag = acc[r];
    }
    lsum += p[r];
  }
  if (PRODUCT) product_from<D, PREC>(tf, p, out);
}

template <int D, int PREC, bool PRODUCT>
__global__ __launch_bounds__(256) void infonce_walk_kernel(Bf16Pack pk, const float* __restrict__ zhat, TermTable T, int n_terms, int B,
                                                           float* __restrict__ lse, float* __restrict__ row_loss, float* __restrict__ dzhat) {
  using Sh = WalkShape<D, PREC, PRODUCT>;
  constexpr int BLK = Sh::BLK, TILE = Sh::TILE, PER = Sh::PER, NS = Sh::NS;
  static_assert(Sh::OK, "shape not supported by the pipelined walk");
  extern __shared__ __attribute__((aligned(16))) __bf16 sm_oth[];  // [NS][TILE]
  typedef __attribute__((address_space(3))) void* lds_p;
  typedef const __attribute__((address_space(1))) void* gbl_p;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  const UnitMap um = xcd_unit((B + 127) >> 7);  // unit = term, sub = block of 128 own rows
  if (um.unit >= n_terms) return;
  const int t = um.unit;
  const int o0 = (um.sub * 4 + wave) * 32;
  const bool active = o0 < B;
  const dmm_infonce_term term = T.t[t];
  const int own_slot = term.a, oth_slot = term.b;
  const int tiles = pk.Bp >> 5;
  const size_t oth_base = (size_t)oth_slot * tiles * BLK;
  const __bf16* srcb[PER];  // wave-uniform source of piece i (see infonce_bwd_lds_kernel)
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c0 = i * 256 + wave * 64, blk = c0 / (BLK / 8);
    const __bf16* base = PRODUCT ? (blk == 0 ? pk.hi : blk == 1 ? pk.hiT : blk == 2 ? pk.lo : pk.loT) : (blk == 0 ? pk.hi : pk.lo);
    srcb[i] = base + oth_base + (size_t)(c0 % (BLK / 8)) * 8;
  }
  auto issue = [&](int kt, int stage) {
    const int k = kt < tiles ? kt : tiles - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds((gbl_p)(srcb[i] + (size_t)k * BLK + (size_t)lane * 8),
                                       (lds_p)(sm_oth + (size_t)stage * TILE + (size_t)(i * 256 + wave * 64) * 8), 16, 0, 0);
  };
  bf16x8 own_h[D / 16], own_l[D / 16];
  const int otile = active ? (o0 >> 5) : 0;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const size_t o = frag_off<D>(own_slot, tiles, otile, ks, fh, fr);
    own_h[ks] = *reinterpret_cast<const bf16x8*>(pk.hi + o);
    if (PREC == 3) own_l[ks] = *reinterpret_cast<const bf16x8*>(pk.lo + o);
  }
  f32x16 out[D / 32];
#pragma unroll
  for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[cb][r] = 0.f;
  const float inv_t = 1.0f / term.temp;
  const float c1 = inv_t * 1.4426950408889634f;
  const int oidx = o0 + fr;
  float lsum = 0.f, diag = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NS - 1; ++j) issue(j, j);
  // the pipelined walk of infonce_bwd_lds_kernel: a step issues the logits of tile kt + 1, then works on tile kt's
  int kt = 0;
  f32x16 acc_a, acc_b;
  inb_wait_vmcnt<(NS - 2) * PER>();
  asm volatile("s_barrier" ::: "memory");
  acc_a = tile_logits<D, PREC, Sh::LO>(sm_oth, own_h, own_l, fr, fh);
  constexpr bool EARLY = D <= 64;
  auto step = [&](auto checked, f32x16& cur, f32x16& nxt) {
    inb_wait_vmcnt<(NS - 3) * PER>();
    asm volatile("s_barrier" ::: "memory");
    issue(kt + NS - 1, (kt + NS - 1) % NS);
    // every LDS read of the step up front (see infonce_bwd_lds_kernel)
    const SFrags<D, PREC> sf = load_sfrags<D, PREC, Sh::LO>(sm_oth + (size_t)((kt + 1) % NS) * TILE, fr, fh);
    TFrags<D, PREC> tf;
    if (PRODUCT) tf = load_tfrags<D, PREC>(sm_oth + (size_t)(kt % NS) * TILE, fr, fh);
    if (EARLY) __builtin_amdgcn_sched_barrier(0);
    nxt = logits_from<D, PREC>(sf, own_h, own_l);
    __builtin_amdgcn_sched_barrier(0);
    if (active) inf_apply<D, PREC, decltype(checked)::value, PRODUCT>(tf, cur, out, kt * 32, oidx, B, c1, lsum, diag, fh);
    ++kt;
  };
#pragma unroll 1
  for (int phase = 0; phase < 2; ++phase) {  // (separate loops for the plain and the checked tiles: see infonce_bwd_lds_kernel)
    const int special = phase == 0 ? otile : tiles - 1;
    const int upto = special < kt ? kt : special;
#pragma unroll 1
    while (kt + 1 < upto) {
      step(std::false_type{}, acc_a, acc_b);
      step(std::false_type{}, acc_b, acc_a);
    }
    if (kt < upto) {
      step(std::false_type{}, acc_a, acc_b);
      acc_a = acc_b;
    }
    if (kt < tiles) {
      step(std::true_type{}, acc_a, acc_b);
      acc_a = acc_b;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!active) return;
  // the two lane halves hold disjoint other-index subsets of the same own row
  const float L = lsum + __shfl_xor(lsum, 32, 64);
  const float dg = diag + __shfl_xor(diag, 32, 64);
  const float l = (c1 + __log2f(L)) * 0.6931471805599453f;
  if (fh == 0 && oidx < B) {
    lse[(size_t)t * B + oidx] = l;
    row_loss[(size_t)t * B + oidx] = (l - dg * inv_t) * (term.weight / (float)B);
  }
  if (!PRODUCT) return;
  const float invL = 1.0f / L;
  const float coef = term.weight * inv_t / (float)B;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rho = (r & 3) + 8 * (r >> 2) + 4 * fh;  // the own row this accumulator register belongs to
    const float il = __shfl(invL, rho, 64);
    const int row = o0 + rho;
    if (row < B) {
#pragma unroll
      for (int cb = 0; cb < D / 32; ++cb) {
        float other = zhat[((size_t)oth_slot * B + row) * D + cb * 32 + fr];  // the positive pair: row `row` of the other view
        if (PREC == 1) other = (float)(__bf16)other;                         // (what the MFMA path of the two-pass backward subtracts)
        atomicAdd(dzhat + ((size_t)own_slot * B + row) * D + cb * 32 + fr, (out[cb][r] * il - other) * coef);
      }
    }
  }
}

// PRODUCT = true: forward + anchor-side backward; false: forward alone (dzhat unused)
template <int D, int PREC, bool PRODUCT>
static void walk_launch(int n_terms, hipStream_t s, Bf16Pack pk, const float* zhat, const TermTable& T, int B, float* lse, float* row_loss, float* dzhat) {
  if constexpr (WalkShape<D, PREC, PRODUCT>::OK) {
    static const bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(infonce_walk_kernel<D, PREC, PRODUCT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                WalkShape<D, PREC, PRODUCT>::LDS_BYTES);
      return true;
    }();
    (void)once;
    constexpr size_t lds_bytes = WalkShape<D, PREC, PRODUCT>::LDS_BYTES;
    hipLaunchKernelGGL((infonce_walk_kernel<D, PREC, PRODUCT>), dim3(xcd_grid(n_terms, cdiv(B, 128))), dim3(256), lds_bytes, s, pk, zhat, T, n_terms, B, lse,
                       row_loss, dzhat);
  }
}
static bool walk_ok(int D, int precision, bool product) {  // WalkShape<>::OK of the instantiations the dispatch below can reach
  if (product) return !(D == 256 && precision == 3);
  return D >= 64;
}

// ---------------------------------------------------------------------------------------------
// BPR forward + backward
// ---------------------------------------------------------------------------------------------
// user rows come from fin_u (indexed by users[b]), item rows from fin_i (indexed by pos[b] / neg[b]): one [U + I, D] table in the
// unsharded engine, the exchanged batch table + the replicated item rows in the user-sharded one
template <int D>
__global__ __launch_bounds__(256) void bpr_kernel(const float* __restrict__ fin_u, const float* __restrict__ fin_i, const int32_t* __restrict__ users,
                                                   const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int B,
                                                   float* __restrict__ row_loss, float* __restrict__ dfin_u, float* __restrict__ dfin_i) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int l = threadIdx.x % LANES;
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < B; b += gridDim.x * G) {
    // a lane owns columns l + j LANES (j = 0..3): each load and each atomic instruction covers one contiguous segment per row (see
    // scatter_normalize_bwd_kernel)
    const size_t ou = (size_t)users[b] * D + l, op = (size_t)pos[b] * D + l, on = (size_t)neg[b] * D + l;
    float u[4], p[4], n[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { u[j] = fin_u[ou + j * LANES]; p[j] = fin_i[op + j * LANES]; n[j] = fin_i[on + j * LANES]; }
    const float sp = group_sum<LANES>(u[0] * p[0] + u[1] * p[1] + u[2] * p[2] + u[3] * p[3]);
    const float sn = group_sum<LANES>(u[0] * n[0] + u[1] * n[1] + u[2] * n[2] + u[3] * n[3]);
    const float sg = 1.0f / (1.0f + expf(-(sp - sn)));
    if (l == 0) row_loss[b] = -logf(1e-5f + sg) * invB;
    if (dfin_u) {
      const float g = -(sg * (1.0f - sg)) / (1e-5f + sg) * invB;  // d row_loss / d (sp - sn)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dfin_u + ou + j * LANES, g * (p[j] - n[j]));
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dfin_i + op + j * LANES, g * u[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dfin_i + on + j * LANES, -g * u[j]);
    }
  }
}

static int fill_slots(const dmm_gather_desc* slots, int n, SlotTable& S) {
  if (!slots || n < 1 || n > 16) {
    set_error("slot table: n_slots=%d must be 1..16", n);
    return DMM_ERR_INVALID_ARG;
  }
  memset(&S, 0, sizeof(S));
  for (int i = 0; i < n; ++i) {
    if (!slots[i].src || !slots[i].idx) {
      set_error("slot %d: null src/idx", i);
      return DMM_ERR_INVALID_ARG;
    }
    S.src[i] = slots[i].src;
    S.idx[i] = slots[i].idx;
    S.grad[i] = slots[i].grad;
  }
  return 0;
}
static int fill_terms(const dmm_infonce_term* terms, int n, TermTable& T) {
  if (!terms || n < 1 || n > 16) {
    set_error("term table: n_terms=%d must be 1..16", n);
    return DMM_ERR_INVALID_ARG;
  }
  memset(&T, 0, sizeof(T));
  for (int i = 0; i < n; ++i) {
    if (terms[i].a < 0 || terms[i].a >= 16 || terms[i].b < 0 || terms[i].b >= 16 || !(terms[i].temp > 0.f)) {
      set_error("term %d: bad slot id or temperature", i);
      return DMM_ERR_INVALID_ARG;
    }
    T.t[i] = terms[i];
  }
  return 0;
}
static bool d_ok(int D) { return D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256; }

#define DISPATCH_D(D, CALL)                                                   \
  switch (D) {                                                                \
    case 8: { constexpr int DD = 8; CALL; } break;                            \
    case 16: { constexpr int DD = 16; CALL; } break;                          \
    case 32: { constexpr int DD = 32; CALL; } break;                          \
    case 64: { constexpr int DD = 64; CALL; } break;                          \
    case 128: { constexpr int DD = 128; CALL; } break;                        \
    case 256: { constexpr int DD = 256; CALL; } break;                        \
  }

}  // namespace dmm
using namespace dmm;

static Bf16Pack make_pack(void* base, int n_slots, int B, int D) {
  Bf16Pack pk = {nullptr, nullptr, nullptr, nullptr, (B + 31) & ~31};
  if (base) {
    const size_t n = (size_t)n_slots * pk.Bp * D;
    pk.hi = (__bf16*)base;
    pk.lo = pk.hi + n;
    pk.hiT = pk.lo + n;
    pk.loT = pk.hiT + n;
  }
  return pk;
}
static bool mfma_ok(int D) { return D == 32 || D == 64 || D == 128 || D == 256; }

extern "C" int64_t dmm_infonce_pack_bytes(int n_slots, int B, int D) { return (int64_t)4 * n_slots * ((B + 31) & ~31) * D * 2; }

static int gather_normalize(const dmm_gather_desc* slots, int n_slots, int B, int D, float* zhat, float* norm, void* bf16_pack, bool lo, void* stream);
extern "C" int dmm_gather_normalize(const dmm_gather_desc* slots, int n_slots, int B, int D, float* zhat, float* norm, void* bf16_pack,
                                    void* stream) {
  return gather_normalize(slots, n_slots, B, D, zhat, norm, bf16_pack, true, stream);
}
extern "C" int dmm_gather_normalize_hi(const dmm_gather_desc* slots, int n_slots, int B, int D, float* zhat, float* norm, void* bf16_pack,
                                       void* stream) {
  return gather_normalize(slots, n_slots, B, D, zhat, norm, bf16_pack, false, stream);
}
static int gather_normalize(const dmm_gather_desc* slots, int n_slots, int B, int D, float* zhat, float* norm, void* bf16_pack, bool lo, void* stream) {
  SlotTable S;
  if (int e = fill_slots(slots, n_slots, S)) return e;
  DMM_CHECK_ARG(zhat && norm && B > 0 && d_ok(D), "dmm_gather_normalize: bad arguments (B=%d D=%d)", B, D);
  DMM_CHECK_ARG(!bf16_pack || (((uintptr_t)bf16_pack & 15) == 0), "dmm_gather_normalize: bf16 pack must be 16-byte aligned");
  const Bf16Pack pk = make_pack(bf16_pack, n_slots, B, D);
  const int G = 1024 / D;
  const int rows = bf16_pack ? pk.Bp : B;
  dim3 grid(cdiv(rows, G) > 1024 ? 1024 : cdiv(rows, G), n_slots);
  if (lo || !bf16_pack) { DISPATCH_D(D, hipLaunchKernelGGL(gather_normalize_kernel<DD>, grid, dim3(256), 0, (hipStream_t)stream, S, B, zhat, norm, pk)); }
  else { DISPATCH_D(D, hipLaunchKernelGGL((gather_normalize_kernel<DD, false>), grid, dim3(256), 0, (hipStream_t)stream, S, B, zhat, norm, pk)); }
  DMM_LAUNCH_CHECK("dmm_gather_normalize");
  return 0;
}

extern "C" int dmm_scatter_normalize_bwd(const dmm_gather_desc* slots, int n_slots, int B, int D, const float* zhat, const float* norm,
                                         const float* dzhat, void* stream) {
  SlotTable S;
  if (int e = fill_slots(slots, n_slots, S)) return e;
  DMM_CHECK_ARG(zhat && norm && dzhat && B > 0 && d_ok(D), "dmm_scatter_normalize_bwd: bad arguments");
  const int G = 1024 / D;
  dim3 grid(cdiv(B, G) > 1024 ? 1024 : cdiv(B, G), n_slots);
  DISPATCH_D(D, hipLaunchKernelGGL(scatter_normalize_bwd_kernel<DD>, grid, dim3(256), 0, (hipStream_t)stream, S, B, zhat, norm, dzhat));
  DMM_LAUNCH_CHECK("dmm_scatter_normalize_bwd");
  return 0;
}

static int infonce_waves(int B, int D, bool bwd) {
  const size_t per_wave = (size_t)((D + (bwd ? B : 0) + 3) & ~3) * sizeof(float);
  for (int w = 4; w >= 1; w >>= 1)
    if (per_wave * w <= 60 * 1024) return w;
  return 0;
}

#define DISPATCH_MFMA(D, PREC, KERNEL, ...)                                                                    \
  switch (D) {                                                                                                 \
    case 32: if (PREC == 3) hipLaunchKernelGGL((KERNEL<32, 3>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<32, 1>), __VA_ARGS__); break;     \
    case 64: if (PREC == 3) hipLaunchKernelGGL((KERNEL<64, 3>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<64, 1>), __VA_ARGS__); break;     \
    case 128: if (PREC == 3) hipLaunchKernelGGL((KERNEL<128, 3>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<128, 1>), __VA_ARGS__); break; \
    case 256: if (PREC == 3) hipLaunchKernelGGL((KERNEL<256, 3>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<256, 1>), __VA_ARGS__); break; \
  }

extern "C" int dmm_infonce_fwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const dmm_infonce_term* terms, int n_terms,
                               int B, int D, float* lse, float* row_loss, void* stream) {
  TermTable T;
  if (int e = fill_terms(terms, n_terms, T)) return e;
  DMM_CHECK_ARG(zhat && lse && row_loss && B > 0 && D % 4 == 0 && D > 0, "dmm_infonce_fwd: bad arguments");
  if (bf16_pack && precision != 0 && mfma_ok(D)) {
    DMM_CHECK_ARG(precision == 1 || precision == 3, "dmm_infonce_fwd: precision must be 0 (fp32 VALU), 1 or 3");
    const Bf16Pack pk = make_pack(const_cast<void*>(bf16_pack), n_slots, B, D);
    DISPATCH_MFMA(D, precision, infonce_fwd_mfma_kernel, dim3(xcd_grid(n_terms, cdiv(B, 32))), dim3(256), 0, (hipStream_t)stream, pk, T, n_terms, B, lse,
                  row_loss);
    DMM_LAUNCH_CHECK("dmm_infonce_fwd(mfma)");
    return 0;
  }
  const int w = infonce_waves(B, D, false);
  DMM_CHECK_ARG(w > 0, "dmm_infonce_fwd: D=%d too large", D);
  hipLaunchKernelGGL(infonce_fwd_kernel, dim3(cdiv(B, w), n_terms), dim3(64 * w), (size_t)w * D * sizeof(float), (hipStream_t)stream,
                     zhat, T, B, D, lse, row_loss);
  DMM_LAUNCH_CHECK("dmm_infonce_fwd");
  return 0;
}

extern "C" int dmm_infonce_bwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const float* lse,
                               const dmm_infonce_term* terms, int n_terms, int B, int D, float* dzhat, void* stream) {
  TermTable T;
  if (int e = fill_terms(terms, n_terms, T)) return e;
  DMM_CHECK_ARG(zhat && lse && dzhat && B > 0 && D % 4 == 0 && D > 0, "dmm_infonce_bwd: bad arguments");
  if (bf16_pack && precision != 0 && mfma_ok(D)) {
    DMM_CHECK_ARG(precision == 1 || precision == 3, "dmm_infonce_bwd: precision must be 0 (fp32 VALU), 1 or 3");
    const Bf16Pack pk = make_pack(const_cast<void*>(bf16_pack), n_slots, B, D);
    static const int ks_env = [] { const char* e = getenv("DMM_INFONCE_KSPLIT"); return e ? atoi(e) : 0; }();  // tuning knob: > 0 = the per-wave walk
    if (ks_env <= 0 && B <= INB_MAX_B) {  // default: the LDS-shared walk
      hipStream_t st = (hipStream_t)stream;
      switch (D) {
        case 32: if (precision == 3) inb_launch<32, 3>(n_terms, 2, st, pk, lse, T, B, dzhat); else inb_launch<32, 1>(n_terms, 2, st, pk, lse, T, B, dzhat); break;
        case 64: if (precision == 3) inb_launch<64, 3>(n_terms, 2, st, pk, lse, T, B, dzhat); else inb_launch<64, 1>(n_terms, 2, st, pk, lse, T, B, dzhat); break;
        case 128: if (precision == 3) inb_launch<128, 3>(n_terms, 2, st, pk, lse, T, B, dzhat); else inb_launch<128, 1>(n_terms, 2, st, pk, lse, T, B, dzhat); break;
        case 256: if (precision == 3) inb_launch<256, 3>(n_terms, 2, st, pk, lse, T, B, dzhat); else inb_launch<256, 1>(n_terms, 2, st, pk, lse, T, B, dzhat); break;
      }
      DMM_LAUNCH_CHECK("dmm_infonce_bwd(mfma, lds)");
      return 0;
    }
    const int ksplit = ks_env > 0 ? ks_env : INFONCE_KSPLIT;
    DISPATCH_MFMA(D, precision, infonce_bwd_mfma_kernel, dim3(cdiv(B, 128), n_terms, 2 * ksplit), dim3(256), 0, (hipStream_t)stream, pk, lse, T, B, dzhat);
    DMM_LAUNCH_CHECK("dmm_infonce_bwd(mfma)");
    return 0;
  }
  const int w = infonce_waves(B, D, true);
  DMM_CHECK_ARG(w > 0, "dmm_infonce_bwd: batch %d too large for the v1 kernel (LDS row buffer)", B);
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3(cdiv(B, w), n_terms, 2), dim3(64 * w), (size_t)w * ((D + B + 3) & ~3) * sizeof(float),
                     (hipStream_t)stream, zhat, lse, T, B, D, dzhat);
  DMM_LAUNCH_CHECK("dmm_infonce_bwd");
  return 0;
}

extern "C" int dmm_infonce_fwd_bwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const dmm_infonce_term* terms, int n_terms,
                                   int B, int D, float* lse, float* row_loss, float* dzhat, void* stream) {
  TermTable T;
  if (int e = fill_terms(terms, n_terms, T)) return e;
  DMM_CHECK_ARG(zhat && lse && row_loss && dzhat && B > 0 && D % 4 == 0 && D > 0, "dmm_infonce_fwd_bwd: bad arguments");
  bool unit = bf16_pack && (precision == 1 || precision == 3) && mfma_ok(D) && B <= INB_MAX_B;
  for (int i = 0; i < n_terms && unit; ++i) unit = terms[i].temp >= 0.024f;  // exp2((s - 1) log2(e) / temp) must not underflow for s in [-1, 1]
  // Three forms (B = 4096, d = 64, 8 terms, round 4; kernel times under rocprofv3):
  //  * default: the per-wave forward (running maximum: any temperature, rows of any norm) 37 us + the two-pass backward 88-92 us;
  //  * DMM_INFONCE_FUSED=1: one walk gives lse, the row loss and the anchor-side gradient (75 us: 256 workgroups = one wave per SIMD), the backward
  //    keeps pass 1 only (~50 us) -- the same total, so the default stays the form whose forward is bit-identical to dmm_infonce_fwd's;
  //  * DMM_INFONCE_WALK_FWD=1: the walk as forward alone (48 us, again one wave per SIMD) + the two-pass backward.
  static const int allow_fused = [] { const char* e = getenv("DMM_INFONCE_FUSED"); return e ? atoi(e) : 0; }();
  static const int allow_walk = [] { const char* e = getenv("DMM_INFONCE_WALK_FWD"); return e ? atoi(e) : 0; }();
  const bool fused = unit && allow_fused && walk_ok(D, precision, true);
  const bool walk_fwd = unit && !fused && allow_walk && walk_ok(D, precision, false);
  if (!fused && !walk_fwd) {  // the two-kernel path (any arithmetic mode, any temperature)
    if (int e = dmm_infonce_fwd(zhat, bf16_pack, n_slots, precision, terms, n_terms, B, D, lse, row_loss, stream)) return e;
    return dmm_infonce_bwd(zhat, bf16_pack, n_slots, precision, lse, terms, n_terms, B, D, dzhat, stream);
  }
  const Bf16Pack pk = make_pack(const_cast<void*>(bf16_pack), n_slots, B, D);
  hipStream_t st = (hipStream_t)stream;
#define DMM_WALK(DD, PP)                                                                       \
  if (fused) {                                                                                 \
    walk_launch<DD, PP, true>(n_terms, st, pk, zhat, T, B, lse, row_loss, dzhat);              \
    inb_launch<DD, PP>(n_terms, 1, st, pk, lse, T, B, dzhat, 1);                               \
  } else {                                                                                     \
    walk_launch<DD, PP, false>(n_terms, st, pk, zhat, T, B, lse, row_loss, dzhat);             \
    inb_launch<DD, PP>(n_terms, 2, st, pk, lse, T, B, dzhat);                                  \
  }
  switch (D) {
    case 32: if (precision == 3) { DMM_WALK(32, 3) } else { DMM_WALK(32, 1) } break;
    case 64: if (precision == 3) { DMM_WALK(64, 3) } else { DMM_WALK(64, 1) } break;
    case 128: if (precision == 3) { DMM_WALK(128, 3) } else { DMM_WALK(128, 1) } break;
    case 256: if (precision == 3) { DMM_WALK(256, 3) } else { DMM_WALK(256, 1) } break;
  }
#undef DMM_WALK
  DMM_LAUNCH_CHECK("dmm_infonce_fwd_bwd");
  return 0;
}


extern "C" int dmm_bpr_fwd_bwd_tables(const float* fin_u, const float* fin_i, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int D,
                                      float* row_loss, float* dfin_u, float* dfin_i, void* stream) {
  DMM_CHECK_ARG(fin_u && fin_i && users && pos && neg && row_loss && B > 0 && d_ok(D), "dmm_bpr_fwd_bwd: bad arguments");
  DMM_CHECK_ARG((dfin_u == nullptr) == (dfin_i == nullptr), "dmm_bpr_fwd_bwd: give both gradient tables or none");
  const int G = 1024 / D;
  const unsigned grid = cdiv(B, G) > 1024 ? 1024 : cdiv(B, G);
  DISPATCH_D(D, hipLaunchKernelGGL(bpr_kernel<DD>, dim3(grid), dim3(256), 0, (hipStream_t)stream, fin_u, fin_i, users, pos, neg, B, row_loss, dfin_u, dfin_i));
  DMM_LAUNCH_CHECK("dmm_bpr_fwd_bwd");
  return 0;
}
extern "C" int dmm_bpr_fwd_bwd(const float* fin, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int D, int U,
                               float* row_loss, float* dfin, void* stream) {
  DMM_CHECK_ARG(fin && U >= 0, "dmm_bpr_fwd_bwd: bad arguments");
  return dmm_bpr_fwd_bwd_tables(fin, fin + (size_t)U * D, users, pos, neg, B, D, row_loss, dfin, dfin ? dfin + (size_t)U * D : nullptr, stream);
}

static int rows_move(bool scatter, const dmm_gather_desc* slots, int n_slots, int B, int D, const int32_t* seg, float* buf, void* stream) {
  SlotTable S;
  if (int e = fill_slots(slots, n_slots, S)) return e;
  DMM_CHECK_ARG(buf && B > 0 && d_ok(D), "dmm_gather_rows / dmm_scatter_add_rows: bad arguments (B=%d D=%d)", B, D);
  const int G = 1024 / D;
  dim3 grid(cdiv(B, G) > 1024 ? 1024 : cdiv(B, G), n_slots);
  if (scatter) {
    DISPATCH_D(D, hipLaunchKernelGGL((rows_move_kernel<DD, true>), grid, dim3(256), 0, (hipStream_t)stream, S, B, seg, buf));
  } else {
    DISPATCH_D(D, hipLaunchKernelGGL((rows_move_kernel<DD, false>), grid, dim3(256), 0, (hipStream_t)stream, S, B, seg, buf));
  }
  DMM_LAUNCH_CHECK("dmm_gather_rows / dmm_scatter_add_rows");
  return 0;
}
extern "C" int dmm_gather_rows(const dmm_gather_desc* slots, int n_slots, int B, int D, const int32_t* seg, float* out, void* stream) {
  return rows_move(false, slots, n_slots, B, D, seg, out, stream);
}
extern "C" int dmm_scatter_add_rows(const dmm_gather_desc* slots, int n_slots, int B, int D, const int32_t* seg, const float* in, void* stream) {
  return rows_move(true, slots, n_slots, B, D, seg, const_cast<float*>(in), stream);
}
