// Products between an item-space row block [B, I] and an [I, d] feature / embedding table in the diffusion training step -- shapes a
// square-tile GEMM handles badly (one side is d = 64 or 128) and that are bound by streaming the [B, I] operand once:
//   dmm_item_project   out[B, d] += X[B, :I] . F[I, d]      Denoise gate input  x_t @ modal_feat            (Model.py:207)
//                                                           usr_model_embeds = model_output @ model_feats   (Model.py:413)
//                                                           usr_id_embeds    = x_start @ itmEmbeds          (Model.py:414)
//                                                           and the gradient  d(proj * gate) = dx' @ modal_feat
//   dmm_item_expand    Y[B, :I] = R[B, :I] + P[B, d] . F[I, d]^T   x_t + (proj * gate) @ modal_feat.T      (Model.py:209)
//                                                           and the gradient  dout += dum @ model_feats.T
// Arithmetic: operands rounded to bf16 as they are loaded (the "precision 1" mode of dmm_gemm), MFMA 16x16x32, fp32 accumulation.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace dmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 is_cvt8(const float4& a, const float4& b) {
  return bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}

// ---- project: (64 MT) rows x D per workgroup (4 waves x 16 MT rows), one K slice of `kchunk` items; F is staged transposed in LDS -------
// MT = 4: the [B, I] x [I, d] products of the diffusion step (few, long rows of X).  MT = 1: the modality-feature projections of the joint
// step (Model.py:84-85,103: many short rows), where FT says F is given as [D, K] (an nn.Linear weight) and OUT_T stores the result
// transposed (the weight gradient dW = dfeat^T E as  E^T[F, I] x dfeat[I, D]  written into the [D, F] gradient).  Up to 4 problems per launch.
constexpr int IP_KS = 128;  // items per LDS stage
struct ProjP {
  const float* X; long long ldx; const float* F; long long ldf; float* out; long long ldo; const float* bias; int rows, K, kchunk;
};
struct ProjBatch { ProjP p[4]; };
struct ProjOne { ProjP p[1]; };  // (the single-problem launch: a 72-byte kernel argument)

// XB (MT == 1 only): X is given in bf16 ([rows, ldx] bf16 elements) -- e.g. a transposed bf16 weight copy that exists anyway
template <int D, int MT, bool FT, bool OUT_T, typename BatchT, bool XB = false>
__global__ __launch_bounds__(256) void item_project_kernel(const BatchT batch) {
  static_assert(!XB || MT == 1, "bf16 X: 64-row workgroups only");
  constexpr int NT = D / 16;
  constexpr int KSG = D > 128 ? IP_KS / 2 : IP_KS;  // items per LDS stage (d = 256: half, so that D feature columns still fit 64 KiB)
  constexpr int NSTEP = KSG / 32;
  constexpr int LDT = KSG * 2 + 16;  // bytes per feature column in LDS (+16: the 16 columns of a fragment read hit distinct banks)
  __shared__ __attribute__((aligned(16))) char ft[D * LDT];
  const ProjP& P = batch.p[blockIdx.z];
  const float* __restrict__ X = P.X; const float* __restrict__ F = P.F; float* __restrict__ out = P.out;
  const long long ldx = P.ldx, ldf = P.ldf, ldo = P.ldo;
  const int B = P.rows, I = P.K, kchunk = P.kchunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * (64 * MT) + wave * (16 * MT);
  const int kbeg = blockIdx.y * kchunk, kend = min(I, kbeg + kchunk);
  if (kbeg >= kend || blockIdx.x * (64 * MT) >= B) return;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow[mt] = X + (long long)min(row0 + mt * 16 + fr, B - 1) * ldx + kg * 8;
  const __bf16* xrow16 = reinterpret_cast<const __bf16*>(X) + (long long)min(row0 + fr, B - 1) * ldx + kg * 8;  // (XB)

  // ---- MT == 1, d <= 128 (r4): the stages of a workgroup's K slice as a software pipeline.  A stage was: request X, barrier, load F piece by piece
  // into LDS, barrier, 16 MFMAs -- a serial chain of two memory round trips, repeated per stage (the joint step's text modality: 3 stages; the
  // folded products of the diffusion step: dozens).  Here the NEXT stage's X fragments and F pieces are requested into a second register set
  // before the current stage is converted into LDS and multiplied; every load is unconditional (clamped address + select), so the compiler keeps
  // counted waits and the younger set stays in flight across the barriers.  F transposed with K % 4 != 0 (an element-granular tail) keeps the loop below.
  constexpr bool PIPE = MT == 1 && D <= 128;
  if (PIPE && (!FT || (I & 3) == 0)) {
    constexpr int NF = FT ? D * (KSG / 4) / 256 : KSG * (D / 4) / 256;  // 16-byte pieces of the F stage per thread
    struct Stage { float4 xs[NSTEP][2]; float4 fv[NF]; };
    auto request = [&](int k0, Stage& R) {
      const int steps = min(KSG, (kend - k0 + 31) & ~31) / 32;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int kq = k0 + min(s, steps - 1) * 32;
        if (XB) R.xs[s][0] = *reinterpret_cast<const float4*>(xrow16 + kq);
        else {
          R.xs[s][0] = *reinterpret_cast<const float4*>(xrow[0] + kq);
          R.xs[s][1] = *reinterpret_cast<const float4*>(xrow[0] + kq + 4);
        }
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int idx = threadIdx.x + j * 256;
        if (!FT) {  // F is [K, D]: row k of the stage
          const int kk = idx / (D / 4), c4 = idx % (D / 4);
          const int k = min(k0 + kk, kend - 1);
          R.fv[j] = *reinterpret_cast<const float4*>(F + (long long)k * ldf + c4 * 4);
        } else {    // F is [D, K]: four consecutive k of column c
          const int c = idx / (KSG / 4), k4 = (idx % (KSG / 4)) * 4;
          const int k = min(k0 + k4, kend - 4);  // (kend % 4 == 0 here: I % 4 == 0 and whole 32-item slices)
          R.fv[j] = *reinterpret_cast<const float4*>(F + (long long)c * ldf + k);
        }
      }
    };
    auto to_lds = [&](int k0, const Stage& R) {
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int idx = threadIdx.x + j * 256;
        float4 v = R.fv[j];
        if (!FT) {
          const int kk = idx / (D / 4), c4 = idx % (D / 4);
          if (k0 + kk >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 0) * LDT + kk * 2) = (__bf16)v.x;
          *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 1) * LDT + kk * 2) = (__bf16)v.y;
          *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 2) * LDT + kk * 2) = (__bf16)v.z;
          *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 3) * LDT + kk * 2) = (__bf16)v.w;
        } else {
          const int c = idx / (KSG / 4), k4 = (idx % (KSG / 4)) * 4;
          if (k0 + k4 >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
          typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
          *reinterpret_cast<bf16x4v*>(ft + c * LDT + k4 * 2) = bf16x4v{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        }
      }
    };
    auto multiply = [&](int k0, const Stage& R) {
      const int steps = min(KSG, (kend - k0 + 31) & ~31) / 32;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s < steps) {
          const bf16x8 fa = XB ? __builtin_bit_cast(bf16x8, R.xs[s][0]) : is_cvt8(R.xs[s][0], R.xs[s][1]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ft + (nt * 16 + fr) * LDT + (s * 32 + kg * 8) * 2);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[0][nt], 0, 0, 0);
          }
        }
      }
    };
    Stage A, Bq;  // static names: the loop is unrolled by two (runtime-indexed register arrays would go to scratch)
    request(kbeg, A);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * KSG) {
      const bool has1 = k0 + KSG < kend, has2 = k0 + 2 * KSG < kend;
      if (has1) request(k0 + KSG, Bq);
      __syncthreads();  // the previous stage's fragment reads are done
      to_lds(k0, A);
      __syncthreads();
      multiply(k0, A);
      if (!has1) break;
      if (has2) request(k0 + 2 * KSG, A);
      __syncthreads();
      to_lds(k0 + KSG, Bq);
      __syncthreads();
      multiply(k0 + KSG, Bq);
    }
  } else
  for (int k0 = kbeg; k0 < kend; k0 += KSG) {
    const int steps = min(KSG, (kend - k0 + 31) & ~31) / 32;
    // MT == 1 (one 16-row tile per wave: two loads per lane and k-step would leave the loop latency-bound): the X fragments of the WHOLE stage
    // are requested up front, under the staging of F
    float4 xs[MT == 1 ? NSTEP : 1][2];
    if (MT == 1) {
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int kq = k0 + min(s, steps - 1) * 32;
        if (XB) {
          xs[s][0] = *reinterpret_cast<const float4*>(xrow16 + kq);  // 8 bf16 = the MFMA fragment as it is
        } else {
          xs[s][0] = *reinterpret_cast<const float4*>(xrow[0] + kq);
          xs[s][1] = *reinterpret_cast<const float4*>(xrow[0] + kq + 4);
        }
      }
    }
    __syncthreads();  // the previous stage's fragment reads are done
    // stage F[k0 .. k0+KS, :] transposed (column-major, bf16); items past the slice are zero, so whatever X holds there is ignored
    // (requesting a thread's 8 pieces up front instead of one per loop trip was measured: no gain at MT = 4, slower at MT = 1)
    if (!FT) {
      for (int idx = threadIdx.x; idx < KSG * (D / 4); idx += 256) {
        const int kk = idx / (D / 4), c4 = idx % (D / 4);
        const int k = k0 + kk;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) v = *reinterpret_cast<const float4*>(F + (long long)k * ldf + c4 * 4);
        *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 0) * LDT + kk * 2) = (__bf16)v.x;
        *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 1) * LDT + kk * 2) = (__bf16)v.y;
        *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 2) * LDT + kk * 2) = (__bf16)v.z;
        *reinterpret_cast<__bf16*>(ft + (c4 * 4 + 3) * LDT + kk * 2) = (__bf16)v.w;
      }
    } else {  // F is [D, K]: column c of the stage is a contiguous piece of row c
      for (int idx = threadIdx.x; idx < D * (KSG / 4); idx += 256) {
        const int c = idx / (KSG / 4), k4 = (idx % (KSG / 4)) * 4;
        const int k = k0 + k4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* f = F + (long long)c * ldf + k;
        if (k + 3 < kend) v = *reinterpret_cast<const float4*>(f);
        else {
          if (k < kend) v.x = f[0];
          if (k + 1 < kend) v.y = f[1];
          if (k + 2 < kend) v.z = f[2];
        }
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4v*>(ft + c * LDT + k4 * 2) = bf16x4v{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      }
    }
    __syncthreads();
    if (MT == 1) {
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s < steps) {
          const bf16x8 fa = XB ? __builtin_bit_cast(bf16x8, xs[s][0]) : is_cvt8(xs[s][0], xs[s][1]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ft + (nt * 16 + fr) * LDT + (s * 32 + kg * 8) * 2);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[0][nt], 0, 0, 0);
          }
        }
      }
      continue;
    }
    // the X fragments of k-step s+1 are requested before the MFMAs of step s (the request past the last step repeats it: no branch
    // around a load, so the compiler keeps counted waits)
    float4 xa[MT][2], xn[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xa[mt][0] = *reinterpret_cast<const float4*>(xrow[mt] + k0);
      xa[mt][1] = *reinterpret_cast<const float4*>(xrow[mt] + k0 + 4);
    }
    for (int s = 0; s < steps; ++s) {
      const int kn = k0 + min(s + 1, steps - 1) * 32;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        xn[mt][0] = *reinterpret_cast<const float4*>(xrow[mt] + kn);
        xn[mt][1] = *reinterpret_cast<const float4*>(xrow[mt] + kn + 4);
      }
      bf16x8 fb[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(ft + (nt * 16 + fr) * LDT + (s * 32 + kg * 8) * 2);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 fa = is_cvt8(xa[mt][0], xa[mt][1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[nt], acc[mt][nt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        xa[mt][0] = xn[mt][0];
        xa[mt][1] = xn[mt][1];
      }
    }
  }
  if (OUT_T) {
    // transposed result: through LDS, so that consecutive lanes add to consecutive addresses of out[c, row0 ..] (the accumulator layout
    // has consecutive lanes on consecutive COLUMNS: 16 cache lines per atomic instruction instead of 1-2)
    static_assert(!OUT_T || MT == 1, "transposed output: 64-row workgroups only");
    // (d = 256: two halves of 128 columns, so that the tile fits the staging buffer)
    constexpr int TC = D > 128 ? 128 : D, TNT = TC / 16;
    static_assert(TC * 65 * 4 <= D * LDT, "transposed tile must fit the staging buffer");
    float* tile = reinterpret_cast<float*>(ft);  // [TC][65] floats
    const int rbase = blockIdx.x * 64;
#pragma unroll
    for (int h = 0; h < D / TC; ++h) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < TNT; ++nt) tile[(nt * 16 + fr) * 65 + wave * 16 + kg * 4 + r] = acc[0][h * TNT + nt][r];
      __syncthreads();
      for (int idx = threadIdx.x; idx < TC * 64; idx += 256) {
        const int col = idx >> 6, rl = idx & 63;
        if (rbase + rl < B) atomicAdd(out + (long long)(h * TC + col) * ldo + rbase + rl, tile[col * 65 + rl]);
      }
    }
    return;
  }
  // one K slice carries the bias.  It is added to the accumulators up front, unconditionally: a bias value first used inside the row-bounds
  // branches made the compiler wait vmcnt(0) in every branch -- i.e. for the previous row's ATOMICS too (+8 us at B = 4096)
  if (P.bias != nullptr && blockIdx.y == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float bv = P.bias[nt * 16 + fr];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += bv;
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + mt * 16 + kg * 4 + r;
      if (row < B) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) atomicAdd(out + (long long)row * ldo + nt * 16 + fr, acc[mt][nt][r]);
      }
    }
}

// ---- expand: 64 rows x 256 item columns per workgroup; the product is formed transposed (features x rows) so that every lane
// ends up with 4 consecutive item columns of one row: 16-byte residual loads and stores ------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void item_expand_kernel(const float* __restrict__ P, long long ldp, const float* __restrict__ F, long long ldf,
                                                          const float* __restrict__ R, long long ldr, float* __restrict__ Y, long long ldy, int B, int I) {
  constexpr int KS = D / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 256 + wave * 64;
  if (col0 >= I) return;
  const bool vec = (ldr % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)R | (uintptr_t)Y) & 15) == 0;
  // residual first: 16 float4 per lane in flight while the operands are fetched and multiplied
  float4 rr[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = row0 + rt * 16 + fr, col = col0 + ct * 16 + kg * 4;
      rr[ct][rt] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < B && col < I) {
        const float* p = R + (long long)row * ldr + col;
        if (vec && col + 3 < I) rr[ct][rt] = *reinterpret_cast<const float4*>(p);
        else {
          rr[ct][rt].x = p[0];
          if (col + 1 < I) rr[ct][rt].y = p[1];
          if (col + 2 < I) rr[ct][rt].z = p[2];
          if (col + 3 < I) rr[ct][rt].w = p[3];
        }
      }
    }
  // (d = 256: products formed up front, 128 columns of d at a time -- see item_expand_fused_kernel)
  constexpr bool CHUNKED = KS > 4;
  constexpr int KC = CHUNKED ? 4 : KS;
  bf16x8 ff[4][KC], fp[4][KC];
  f32x4 accs[CHUNKED ? 4 : 1][CHUNKED ? 4 : 1];
#pragma unroll
  for (int ch = 0; ch < KS / KC; ++ch) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const float* f = F + (long long)min(col0 + t * 16 + fr, I - 1) * ldf + (ch * KC + ks) * 32 + kg * 8;
        const float* p = P + (long long)min(row0 + t * 16 + fr, B - 1) * ldp + (ch * KC + ks) * 32 + kg * 8;
        ff[t][ks] = is_cvt8(*reinterpret_cast<const float4*>(f), *reinterpret_cast<const float4*>(f + 4));
        fp[t][ks] = is_cvt8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4));
      }
    if (CHUNKED) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          f32x4 a = ch == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accs[ct][rt];
#pragma unroll
          for (int ks = 0; ks < KC; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff[ct][ks], fp[rt][ks], a, 0, 0, 0);
          accs[ct][rt] = a;
        }
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (CHUNKED) acc = accs[CHUNKED ? ct : 0][CHUNKED ? rt : 0];
      else {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff[ct][ks], fp[rt][ks], acc, 0, 0, 0);
      }
      // acc[r] = (item column col0 + ct*16 + kg*4 + r, row row0 + rt*16 + fr)
      const int row = row0 + rt * 16 + fr, col = col0 + ct * 16 + kg * 4;
      if (row < B && col < I) {
        const float4 r4 = rr[ct][rt];
        float* y = Y + (long long)row * ldy + col;
        if (vec && col + 3 < I) *reinterpret_cast<float4*>(y) = make_float4(r4.x + acc[0], r4.y + acc[1], r4.z + acc[2], r4.w + acc[3]);
        else {
          y[0] = r4.x + acc[0];
          if (col + 1 < I) y[1] = r4.y + acc[1];
          if (col + 2 < I) y[2] = r4.z + acc[2];
          if (col + 3 < I) y[3] = r4.w + acc[3];
        }
      }
    }
}

// ---- expand, fused with what consumes it in the diffusion training step -----------------------------------------------------------------
// Same tile and product as item_expand_kernel; the [B, I] result is not written in fp32 at all but as the two bf16 operands the next GEMMs
// read: row-major Y16[B, Kc] and transposed YT[Kt, Bp] (through an LDS tile: 128-byte segments).  Two residual sources:
//   MODE 0 (forward, Model.py:209):   v = R[b, c] + (P F^T)[b, c] for c < I,  v = R[b, c] for I <= c < Kc (the time-embedding / pad columns
//                                     of [x_t | temb | 0]): replaces item_expand + cast + transpose-cast (6 passes over [B, I] -> 3)
//   MODE 1 (backward, Model.py:418-428):  v = gscale w[b] 2/I (out[b, c] - x0[b, c]) + (dum F^T)[b, c] for c < I, 0 beyond -- the gradient of the
//                                     diffusion loss w.r.t. the Denoise output, never materialised in fp32; also its column sums (the
//                                     output layer's bias gradient, fp32 atomics into a zeroed vector): replaces diff_loss(dout) +
//                                     item_expand + transpose-cast + colsum + cast (13 passes -> 4)
// Rows b >= B of the tile (up to Bp, a multiple of 64) are written as zeros into YT.
struct ExpandFusedP {
  const float* P; long long ldp;    // [B, D]
  const float* F; long long ldf;    // [I, D]
  const float* R; long long ldr;    // MODE 0: residual rows; MODE 1: `out` rows
  const float* X0; long long ldx0;  // MODE 1: x_0 rows (dense) ...
  const uint32_t* X0b; long long ldb;  // ... or as bit masks (used when X0 is NULL)
  const double* w; double gscale;   // MODE 1: per-row loss weights
  const double* gs_dev;             // MODE 1: optional device-side factor of gscale
  __bf16* Y16; long long ldy;       // [B, >= Kc]
  __bf16* YT; long long ldt;        // [Kt, >= Bp]
  float* csum;                      // MODE 1: [I]
  int B, I, Kc, Kt;
  const __bf16* F16; long long ldf16;  // optional bf16 copy of F (strip kernel: staged instead of F, half the bytes)
};

template <int D, int MODE>
__global__ __launch_bounds__(256, (D > 128 ? 3 : 1)) void item_expand_fused_kernel(const ExpandFusedP A) {
  constexpr int KS = D / 32;
  (void)sizeof(char[MODE >= 0 && MODE <= 2 ? 1 : -1]);
  constexpr int TLD = 72;  // bf16 elements per column of the transposed tile (64 rows + 8: 16-byte aligned, conflict-poor)
  __shared__ __attribute__((aligned(16))) __bf16 tt[256 * TLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.y * 64, colw = blockIdx.x * 256, col0 = colw + wave * 64;
  const int B = A.B, I = A.I, Kc = A.Kc;
  // Every load is issued up front, UNCONDITIONALLY, on clamped addresses (out-of-range lanes re-read the last row / last 4-column group and
  // are masked when the value is used): a load inside a bounds branch, next to the stores and atomics below, made the compiler wait
  // vmcnt(0) -- i.e. for those stores too -- at every use (27 such waits; loss-gradient mode 127 -> 75 us).  MODE 0: forward; MODE 1: loss
  // gradient with dense x_0 rows; MODE 2: loss gradient with x_0 as bit masks.
  constexpr bool LOSS = MODE != 0;
  float4 rr[4][4];
  float gc[4];
  long long rofs[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int row = row0 + rt * 16 + fr, rc = min(row, B - 1);
    rofs[rt] = (long long)rc;
    gc[rt] = 0.f;
    if (LOSS) {
      const float g = (float)(A.gscale * (A.gs_dev ? *A.gs_dev : 1.0) * A.w[rc] * 2.0 / (double)I);
      gc[rt] = row < B ? g : 0.f;
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int cc = min(col0 + ct * 16 + kg * 4, Kc - 4);  // Kc % 4 == 0, rows readable up to Kc (host check)
      rr[ct][rt] = *reinterpret_cast<const float4*>(A.R + rofs[rt] * A.ldr + cc);
    }
  if (MODE == 2) {  // 16 words per lane: the lane's 4 columns (col % 4 == 0) sit in one word; bits past I are zero
    uint32_t wd[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int cc = min(col0 + ct * 16 + kg * 4, ((I + 31) & ~31) - 4);
        wd[ct][rt] = A.X0b[rofs[rt] * A.ldb + (cc >> 5)] >> (cc & 31);
      }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float4& r = rr[ct][rt];
        const uint32_t w_ = wd[ct][rt];
        // (__fmul_rn: the product is rounded on its own, as when dmm_diff_loss stored it -- no contraction with the add of P F^T below)
        r = make_float4(__fmul_rn(gc[rt], r.x - (float)(w_ & 1u)), __fmul_rn(gc[rt], r.y - (float)((w_ >> 1) & 1u)),
                        __fmul_rn(gc[rt], r.z - (float)((w_ >> 2) & 1u)), __fmul_rn(gc[rt], r.w - (float)((w_ >> 3) & 1u)));
      }
  } else if (MODE == 1) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float4 xx[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int cc = min(col0 + ct * 16 + kg * 4, Kc - 4);  // (x_0 rows share the operand layout: readable up to Kc)
        xx[rt] = *reinterpret_cast<const float4*>(A.X0 + rofs[rt] * A.ldx0 + cc);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float4& r = rr[ct][rt];
        r = make_float4(__fmul_rn(gc[rt], r.x - xx[rt].x), __fmul_rn(gc[rt], r.y - xx[rt].y), __fmul_rn(gc[rt], r.z - xx[rt].z),
                        __fmul_rn(gc[rt], r.w - xx[rt].w));
      }
    }
  }
  // d <= 128: the fragments of the whole d stay in registers and each product is formed where it is used.  d = 256: the products are formed up
  // front, 128 columns of d at a time (the fragments of one half: 128 registers next to the 64 accumulators and the 64 residual registers).
  constexpr bool CHUNKED = KS > 4;
  constexpr int KC = CHUNKED ? 4 : KS;
  bf16x8 ff[4][KC], fp[4][KC];
  f32x4 accs[CHUNKED ? 4 : 1][CHUNKED ? 4 : 1];
#pragma unroll
  for (int ch = 0; ch < KS / KC; ++ch) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const float* f = A.F + (long long)min(col0 + t * 16 + fr, I - 1) * A.ldf + (ch * KC + ks) * 32 + kg * 8;
        const float* p = A.P + (long long)min(row0 + t * 16 + fr, B - 1) * A.ldp + (ch * KC + ks) * 32 + kg * 8;
        ff[t][ks] = is_cvt8(*reinterpret_cast<const float4*>(f), *reinterpret_cast<const float4*>(f + 4));
        fp[t][ks] = is_cvt8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4));
      }
    if (CHUNKED) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          f32x4 a = ch == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accs[ct][rt];
#pragma unroll
          for (int ks = 0; ks < KC; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff[ct][ks], fp[rt][ks], a, 0, 0, 0);
          accs[ct][rt] = a;
        }
    }
  }
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (CHUNKED) acc = accs[CHUNKED ? ct : 0][CHUNKED ? rt : 0];
      else {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff[ct][ks], fp[rt][ks], acc, 0, 0, 0);
      }
      // acc[r] = (item column col0 + ct*16 + kg*4 + r, row row0 + rt*16 + fr)
      const int row = row0 + rt * 16 + fr, col = col0 + ct * 16 + kg * 4;
      const float4 r4 = rr[ct][rt];
      float v[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (LOSS) {  // loss gradient: defined on the item columns only
          v[j] = col + j < I ? v[j] + acc[j] : 0.f;
        } else {     // forward: [x_t | temb | 0] beyond the item columns is the residual itself; nothing beyond Kc is stored
          if (col + j < I) v[j] += acc[j];
        }
        if (row >= B || col >= Kc) v[j] = 0.f;
        cs[j] += v[j];
      }
      const bf16x4v b4 = bf16x4v{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      if (row < B && col < Kc) {  // Kc % 4 == 0 and 8-byte aligned rows (checked on the host)
        *reinterpret_cast<bf16x4v*>(A.Y16 + (long long)row * A.ldy + col) = b4;
      }
      const int cl = wave * 64 + ct * 16 + kg * 4, rl = rt * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) tt[(cl + j) * TLD + rl] = b4[j];
    }
    if (LOSS) {  // column sums over the tile's 64 rows: reduce over fr (the 16 lanes of a kg group), then one atomic per column
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float c = cs[j];
        c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64); c += __shfl_xor(c, 8, 64);
        const int col = col0 + ct * 16 + kg * 4 + j;
        if (fr == 0 && col < I) atomicAdd(A.csum + col, c);
      }
    }
  }
  __syncthreads();
  // transposed copy: thread t owns tile column t = 64 consecutive batch rows = 128 bytes of YT's row colw + t
  const int c = colw + threadIdx.x;
  if (c < A.Kt) {
    const uint4* src = reinterpret_cast<const uint4*>(tt + threadIdx.x * TLD);
    uint4* dst = reinterpret_cast<uint4*>(A.YT + (long long)c * A.ldt + row0);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = src[i];
  }
}

// ---- the same with the P fragments resident over a strip of item tiles (the default) ---------------------------------------------------------------
// The tile above is bound by its operand fragments, not by the [B, I] passes it fuses: every [64 rows, 256 items] tile loads [256 items, d] of F (fp32,
// each wave its own 64 columns) and, four times over, the [64 rows, d] fragments of P -- at d = 256 that is 16 + 16 bytes of fragment loads from L2 per
// output element against 4 of residual and 4 of output (3.8 / 4.6 ms against a byte floor of 1.7 at 8192 x 100k).  Here a wave keeps the P fragments of
// its 32 rows ([32, d]: 64 registers at d = 256) for the whole kernel and the workgroup (4 waves = 128 rows) walks `tiles` consecutive 64-item tiles:
// per tile it stages the [64, d] slice of F ONCE in LDS as bf16 (row stride d + 8: the 16 rows of a fragment read hit distinct banks), forms the
// products column tile by column tile (8 live accumulators), writes the row-major bf16 result and passes the transposed copy through LDS as
// 256-byte row segments.  Same k order as the tile above: bit-identical results.  (An intermediate form -- [256 rows, 64 items] per workgroup, F in
// LDS, P fragments still per tile -- gained 25-30 % at d = 128 and 10 % at d = 256; this one 35-40 % / 22-37 %.)
template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void item_expand_fused_strip_kernel(const ExpandFusedP A, int tiles) {
  constexpr int KS = D / 32, LDF = D + 8, TLD = 128 + 8;
  __shared__ __attribute__((aligned(16))) __bf16 fl[64 * LDF];
  __shared__ __attribute__((aligned(16))) __bf16 tt[64 * TLD];
  __shared__ float csl[4][64];  // loss modes: the waves' column sums of the current tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.y * 128, rowW = row0 + wave * 32;
  const int B = A.B, I = A.I, Kc = A.Kc;
  constexpr bool LOSS = MODE != 0;
  float gc[2];
  long long rofs[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = rowW + rt * 16 + fr, rc = min(row, B - 1);
    rofs[rt] = (long long)rc;
    gc[rt] = 0.f;
    if (LOSS) {
      const float g = (float)(A.gscale * (A.gs_dev ? *A.gs_dev : 1.0) * A.w[rc] * 2.0 / (double)I);
      gc[rt] = row < B ? g : 0.f;
    }
  }
  bf16x8 fp[2][KS];  // this wave's rows of P, for every tile of the strip
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* p = A.P + rofs[rt] * A.ldp + ks * 32 + kg * 8;
      fp[rt][ks] = is_cvt8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4));
    }
  const int Bp = (B + 63) & ~63;
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  for (int t = 0; t < tiles; ++t) {
    const int col0 = (blockIdx.x * tiles + t) * 64;
    if (col0 >= Kc) break;  // (uniform)
    __syncthreads();        // the previous tile's readers of fl / tt are done
    if (A.F16) {  // (uniform) the caller's bf16 copy of F: the same values, half the bytes
      for (int idx = threadIdx.x; idx < 64 * (D / 8); idx += 256) {
        const int it = idx / (D / 8), c8 = idx % (D / 8);
        *reinterpret_cast<uint4*>(fl + it * LDF + c8 * 8) = *reinterpret_cast<const uint4*>(A.F16 + (long long)min(col0 + it, I - 1) * A.ldf16 + c8 * 8);
      }
    } else {
      for (int idx = threadIdx.x; idx < 64 * (D / 4); idx += 256) {
        const int it = idx / (D / 4), c4 = idx % (D / 4);
        const float4 v = *reinterpret_cast<const float4*>(A.F + (long long)min(col0 + it, I - 1) * A.ldf + c4 * 4);
        __bf16 h[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<uint2*>(fl + it * LDF + c4 * 4) = *reinterpret_cast<const uint2*>(h);
      }
    }
    float4 rr[4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int cc = min(col0 + ct * 16 + kg * 4, Kc - 4);
        rr[ct][rt] = *reinterpret_cast<const float4*>(A.R + rofs[rt] * A.ldr + cc);
      }
    if (MODE == 2) {
      uint32_t wd[4][2];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int cc = min(col0 + ct * 16 + kg * 4, ((I + 31) & ~31) - 4);
          wd[ct][rt] = A.X0b[rofs[rt] * A.ldb + (cc >> 5)] >> (cc & 31);
        }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          float4& r = rr[ct][rt];
          const uint32_t w_ = wd[ct][rt];
          r = make_float4(__fmul_rn(gc[rt], r.x - (float)(w_ & 1u)), __fmul_rn(gc[rt], r.y - (float)((w_ >> 1) & 1u)),
                          __fmul_rn(gc[rt], r.z - (float)((w_ >> 2) & 1u)), __fmul_rn(gc[rt], r.w - (float)((w_ >> 3) & 1u)));
        }
    } else if (MODE == 1) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int cc = min(col0 + ct * 16 + kg * 4, Kc - 4);
          const float4 xx = *reinterpret_cast<const float4*>(A.X0 + rofs[rt] * A.ldx0 + cc);
          float4& r = rr[ct][rt];
          r = make_float4(__fmul_rn(gc[rt], r.x - xx.x), __fmul_rn(gc[rt], r.y - xx.y), __fmul_rn(gc[rt], r.z - xx.z), __fmul_rn(gc[rt], r.w - xx.w));
        }
    }
    __syncthreads();  // F slice staged
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      bf16x8 ff[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) ff[ks] = *reinterpret_cast<const bf16x8*>(fl + (ct * 16 + fr) * LDF + ks * 32 + kg * 8);
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ff[ks], fp[rt][ks], acc, 0, 0, 0);
        // acc[r] = (item column col0 + ct*16 + kg*4 + r, row rowW + rt*16 + fr)
        const int row = rowW + rt * 16 + fr, col = col0 + ct * 16 + kg * 4;
        const float4 r4 = rr[ct][rt];
        float v[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (LOSS) {
            v[j] = col + j < I ? v[j] + acc[j] : 0.f;
          } else {
            if (col + j < I) v[j] += acc[j];
          }
          if (row >= B || col >= Kc) v[j] = 0.f;
          cs[j] += v[j];
        }
        const bf16x4v b4 = bf16x4v{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        if (row < B && col < Kc) *reinterpret_cast<bf16x4v*>(A.Y16 + (long long)row * A.ldy + col) = b4;
        const int cl = ct * 16 + kg * 4, rl = wave * 32 + rt * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) tt[(cl + j) * TLD + rl] = b4[j];
      }
      if (LOSS) {  // column sums over the wave's 32 rows -> LDS; the workgroup adds its 128 rows' sums with ONE atomic instruction per tile, below
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float c = cs[j];
          c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64); c += __shfl_xor(c, 8, 64);
          if (fr == 0) csl[wave][ct * 16 + kg * 4 + j] = c;
        }
      }
    }
    __syncthreads();
    // (64 lanes on 64 consecutive columns: one contiguous 256-byte atomic instruction per workgroup and tile -- per wave and column group it was 64
    // instructions of 4 scattered dwords each, which at 8192 x 100k cost more than the tile's memory traffic: 3.6 ms against 2.4 without the sums)
    if (LOSS && wave == 0 && col0 + lane < I) atomicAdd(A.csum + col0 + lane, (csl[0][lane] + csl[1][lane]) + (csl[2][lane] + csl[3][lane]));
    // transposed copy: thread t owns 32 consecutive batch rows (64 bytes) of tile column t / 4
    const int c = col0 + (int)(threadIdx.x >> 2), q = threadIdx.x & 3;
    if (c < A.Kt && row0 + q * 32 < Bp) {
      const uint4* src = reinterpret_cast<const uint4*>(tt + (threadIdx.x >> 2) * TLD + q * 32);
      uint4* dst = reinterpret_cast<uint4*>(A.YT + (long long)c * A.ldt + row0 + q * 32);
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) dst[i2] = src[i2];
    }
  }
}

}  // namespace dmm
using namespace dmm;

template <int MT, bool FT, bool OUT_T, typename BatchT>
static void project_launch(const BatchT& b, int n, int D, int rb, int ks, hipStream_t s) {
  if (D == 64) hipLaunchKernelGGL((item_project_kernel<64, MT, FT, OUT_T, BatchT>), dim3(rb, ks, n), dim3(256), 0, s, b);
  else if (D == 128) hipLaunchKernelGGL((item_project_kernel<128, MT, FT, OUT_T, BatchT>), dim3(rb, ks, n), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((item_project_kernel<256, (MT > 2 ? 2 : MT), FT, OUT_T, BatchT>), dim3(rb, ks, n), dim3(256), 0, s, b);  // (128 accumulators per lane at most)
}

extern "C" int dmm_item_project(const float* X, int64_t ldx, const float* F, int64_t ldf, float* out, int64_t ldo, int B, int I, int D, void* stream) {
  DMM_CHECK_ARG(X && F && out && B > 0 && I > 0, "dmm_item_project: bad arguments");
  DMM_CHECK_ARG(D == 64 || D == 128 || D == 256, "dmm_item_project: D=%d unsupported (64, 128 or 256)", D);
  DMM_CHECK_ARG(ldx % 4 == 0 && ldf % 4 == 0 && (((uintptr_t)X | (uintptr_t)F) & 15) == 0 && ldx >= ((I + 31) & ~31),
                "dmm_item_project: X and F rows must be 16-byte aligned and X rows readable up to round_up(I, 32)");
  // K slices: enough workgroups to fill the chip a few times over, slices a multiple of 32 items
  const int wg_rows = D == 256 ? 128 : 256;  // rows per workgroup: 4 waves x 16 MT
  const int rb = (B + wg_rows - 1) / wg_rows;
  static const int force_ks = [] { const char* e = getenv("DMM_ITEM_KS"); return e ? atoi(e) : 0; }();  // tuning knob
  int ks = force_ks > 0 ? force_ks : std::max(1, std::min(64, 384 / rb));
  int kchunk = (((I + ks - 1) / ks) + 31) & ~31;
  ks = (I + kchunk - 1) / kchunk;
  ProjOne b;
  b.p[0] = ProjP{X, (long long)ldx, F, (long long)ldf, out, (long long)ldo, nullptr, B, I, kchunk};
  project_launch<4, false, false>(b, 1, D, rb, ks, (hipStream_t)stream);
  DMM_LAUNCH_CHECK("dmm_item_project");
  return 0;
}

static int project_batched(const dmm_project_problem* p, int n, int D, int f_transposed, int out_transposed, bool x_bf16, void* stream);
extern "C" int dmm_item_project_batched(const dmm_project_problem* p, int n, int D, int f_transposed, int out_transposed, void* stream) {
  return project_batched(p, n, D, f_transposed, out_transposed, false, stream);
}
extern "C" int dmm_item_project_batched_x16(const dmm_project_problem* p, int n, int D, void* stream) {
  return project_batched(p, n, D, 0, 1, true, stream);
}
extern "C" int dmm_item_project_batched_xb(const dmm_project_problem* p, int n, int D, int f_transposed, int out_transposed, void* stream) {
  DMM_CHECK_ARG((f_transposed != 0) != (out_transposed != 0), "dmm_item_project_batched_xb: the forms built are (F transposed, out plain) and (F plain, out transposed)");
  return project_batched(p, n, D, f_transposed, out_transposed, true, stream);
}
static int project_batched(const dmm_project_problem* p, int n, int D, int f_transposed, int out_transposed, bool x_bf16, void* stream) {
  DMM_CHECK_ARG(p && n >= 1 && n <= 4, "dmm_item_project_batched: 1..4 problems per launch");
  DMM_CHECK_ARG(D == 64 || D == 128 || D == 256, "dmm_item_project_batched: D=%d unsupported (64, 128 or 256)", D);
  ProjBatch b;
  memset(&b, 0, sizeof(b));
  int rb = 0, ks = 0, rows_all = 0;
  for (int i = 0; i < n; ++i) rows_all += (p[i].rows + 63) / 64;
  for (int i = 0; i < n; ++i) {
    const dmm_project_problem& q = p[i];
    DMM_CHECK_ARG(q.X && q.F && q.out && q.rows > 0 && q.K > 0, "dmm_item_project_batched: bad arguments (problem %d)", i);
    DMM_CHECK_ARG((x_bf16 ? q.ldx % 8 == 0 : q.ldx % 4 == 0) && q.ldf % 4 == 0 && (((uintptr_t)q.X | (uintptr_t)q.F) & 15) == 0 && q.ldx >= ((q.K + 31) & ~31),
                  "dmm_item_project_batched: X and F rows must be 16-byte aligned and X rows readable up to round_up(K, 32) (problem %d)", i);
    DMM_CHECK_ARG(f_transposed ? q.ldf >= q.K : q.ldf >= D, "dmm_item_project_batched: ldf too small (problem %d)", i);
    // K slices of whole 32-item steps: about 1024 workgroups per launch over all problems
    static const int wgs_fwd = [] { const char* e = getenv("DMM_PROJ_WGS"); return e ? atoi(e) : 1024; }();      // tuning knobs
    static const int wgs_bwd = [] { const char* e = getenv("DMM_PROJ_WGS_T"); return e ? atoi(e) : 1024; }();  // (256 / 1024 / 4096 measured: 23.0 / 21.3 / 21.7 us)
    int want = std::max(1, std::min(64, (out_transposed ? wgs_bwd : wgs_fwd) / std::max(1, rows_all)));
    // the forward projections feed replicated state of the user-sharded engine: at most TWO slices, so that the atomic sum is a single
    // commutative addition and the result does not depend on arrival order (bit-identical on every replica)
    static const int fwd_slices = [] { const char* e = getenv("DMM_PROJ_FWD_SLICES"); return e ? atoi(e) : 2; }();  // (experiments: > 2 gives up the bit-identity)
    if (!out_transposed) want = std::min(want, fwd_slices);
    int kchunk = (((q.K + want - 1) / want) + 31) & ~31;
    kchunk = std::max(kchunk, 32);
    b.p[i] = ProjP{q.X, (long long)q.ldx, q.F, (long long)q.ldf, q.out, (long long)q.ldo, q.bias, q.rows, q.K, kchunk};
    rb = std::max(rb, (q.rows + 63) / 64);
    ks = std::max(ks, (q.K + kchunk - 1) / kchunk);
  }
  hipStream_t s = (hipStream_t)stream;
  if (x_bf16 && f_transposed && !out_transposed) {
    if (D == 64) hipLaunchKernelGGL((item_project_kernel<64, 1, true, false, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
    else if (D == 128) hipLaunchKernelGGL((item_project_kernel<128, 1, true, false, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((item_project_kernel<256, 1, true, false, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
  } else if (x_bf16) {
    if (D == 64) hipLaunchKernelGGL((item_project_kernel<64, 1, false, true, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
    else if (D == 128) hipLaunchKernelGGL((item_project_kernel<128, 1, false, true, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((item_project_kernel<256, 1, false, true, ProjBatch, true>), dim3(rb, ks, n), dim3(256), 0, s, b);
  } else if (f_transposed && !out_transposed) project_launch<1, true, false>(b, n, D, rb, ks, s);
  else if (!f_transposed && out_transposed) project_launch<1, false, true>(b, n, D, rb, ks, s);
  else if (!f_transposed && !out_transposed) project_launch<1, false, false>(b, n, D, rb, ks, s);
  else project_launch<1, true, true>(b, n, D, rb, ks, s);
  DMM_LAUNCH_CHECK("dmm_item_project_batched");
  return 0;
}

extern "C" int dmm_item_expand(const float* P, int64_t ldp, const float* F, int64_t ldf, const float* R, int64_t ldr, float* Y, int64_t ldy, int B, int I,
                               int D, void* stream) {
  DMM_CHECK_ARG(P && F && R && Y && B > 0 && I > 0, "dmm_item_expand: bad arguments");
  DMM_CHECK_ARG(D == 64 || D == 128 || D == 256, "dmm_item_expand: D=%d unsupported (64, 128 or 256)", D);
  DMM_CHECK_ARG(ldp % 4 == 0 && ldf % 4 == 0 && (((uintptr_t)P | (uintptr_t)F) & 15) == 0, "dmm_item_expand: P and F rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((I + 255) / 256, (B + 63) / 64);
  if (D == 64) hipLaunchKernelGGL(item_expand_kernel<64>, grid, dim3(256), 0, s, P, (long long)ldp, F, (long long)ldf, R, (long long)ldr, Y, (long long)ldy, B, I);
  else if (D == 128) hipLaunchKernelGGL(item_expand_kernel<128>, grid, dim3(256), 0, s, P, (long long)ldp, F, (long long)ldf, R, (long long)ldr, Y, (long long)ldy, B, I);
  else hipLaunchKernelGGL(item_expand_kernel<256>, grid, dim3(256), 0, s, P, (long long)ldp, F, (long long)ldf, R, (long long)ldr, Y, (long long)ldy, B, I);
  DMM_LAUNCH_CHECK("dmm_item_expand");
  return 0;
}

extern "C" int dmm_item_expand_fused(const dmm_expand_fused_args* a, int mode, int D, void* stream) {
  DMM_CHECK_ARG(a && (mode == 0 || mode == 1), "dmm_item_expand_fused: mode must be 0 (forward) or 1 (loss gradient)");
  DMM_CHECK_ARG(D == 64 || D == 128 || D == 256, "dmm_item_expand_fused: D=%d unsupported (64, 128 or 256)", D);
  DMM_CHECK_ARG(a->P && a->F && a->R && a->Y16 && a->YT && a->B > 0 && a->I > 0 && a->Kc >= a->I && a->Kc % 4 == 0 && a->Kt > 0 && a->Kt <= a->Kc,
                "dmm_item_expand_fused: bad arguments");
  DMM_CHECK_ARG(mode == 0 || ((a->X0 || (a->X0bits && a->ldbits * 32 >= a->I)) && a->w && a->csum), "dmm_item_expand_fused: mode 1 needs x0 (rows or bit masks), w and csum");
  const int Bp = (a->B + 63) & ~63;
  DMM_CHECK_ARG(a->ldp % 4 == 0 && a->ldf % 4 == 0 && a->ldr % 4 == 0 && (mode == 0 || !a->X0 || a->ldx0 % 4 == 0) && a->ldy % 4 == 0 && a->ldy >= a->Kc && a->ldt % 8 == 0 &&
                    a->ldt >= Bp && a->ldr >= a->Kc && (mode == 0 || !a->X0 || a->ldx0 >= a->Kc) && (mode == 0 || a->X0 || a->ldbits * 32 >= ((a->I + 31) & ~31)) &&
                    (((uintptr_t)a->P | (uintptr_t)a->F | (uintptr_t)a->R | (uintptr_t)a->X0 | (uintptr_t)a->YT) & 15) == 0 && ((uintptr_t)a->Y16 & 7) == 0,
                "dmm_item_expand_fused: operands must be 16-byte aligned (Y16: 8) with leading dimensions that are multiples of 4 (YT: of 8, >= round_up(B, 64))");
  ExpandFusedP p;
  p.P = a->P; p.ldp = a->ldp; p.F = a->F; p.ldf = a->ldf; p.R = a->R; p.ldr = a->ldr; p.X0 = a->X0; p.ldx0 = a->ldx0; p.X0b = a->X0bits; p.ldb = a->ldbits; p.w = a->w; p.gscale = a->gscale; p.gs_dev = a->gscale_dev;
  p.Y16 = (__bf16*)a->Y16; p.ldy = a->ldy; p.YT = (__bf16*)a->YT; p.ldt = a->ldt; p.csum = a->csum; p.B = a->B; p.I = a->I; p.Kc = a->Kc; p.Kt = a->Kt;
  p.F16 = (const __bf16*)a->F16; p.ldf16 = a->ldf16;
  DMM_CHECK_ARG(!a->F16 || (a->ldf16 % 8 == 0 && a->ldf16 >= D && ((uintptr_t)a->F16 & 15) == 0), "dmm_item_expand_fused: F16 rows must be 16-byte aligned bf16 [I, >= D]");
  hipStream_t s = (hipStream_t)stream;
  const int km = mode == 0 ? 0 : (a->X0 ? 1 : 2);
  // tile choice (DMM_EXPAND_WIDE, read once): 1 (default) = the strip kernel, 0 = the [64 rows, 256 items] tile
  static const int strip = [] { const char* e = getenv("DMM_EXPAND_WIDE"); return e ? atoi(e) : 1; }();
  if (strip) {
    const int ntile = (a->Kc + 63) / 64, yb = (Bp + 127) / 128;
    int nt = 16;
    while (nt > 1 && (long long)((ntile + nt - 1) / nt) * yb < 1024) nt /= 2;  // enough workgroups to fill the chip four times over
    dim3 gs((ntile + nt - 1) / nt, yb);
#define DMM_EXPAND_STRIP_LAUNCH(DD)                                                                                         \
    do {                                                                                                                      \
      if (km == 0) hipLaunchKernelGGL((item_expand_fused_strip_kernel<DD, 0>), gs, dim3(256), 0, s, p, nt);                 \
      else if (km == 1) hipLaunchKernelGGL((item_expand_fused_strip_kernel<DD, 1>), gs, dim3(256), 0, s, p, nt);            \
      else hipLaunchKernelGGL((item_expand_fused_strip_kernel<DD, 2>), gs, dim3(256), 0, s, p, nt);                         \
    } while (0)
    if (D == 64) DMM_EXPAND_STRIP_LAUNCH(64);
    else if (D == 128) DMM_EXPAND_STRIP_LAUNCH(128);
    else DMM_EXPAND_STRIP_LAUNCH(256);
#undef DMM_EXPAND_STRIP_LAUNCH
    DMM_LAUNCH_CHECK("dmm_item_expand_fused");
    return 0;
  }
  dim3 grid((a->Kc + 255) / 256, Bp / 64);
  if (D == 64) {
    if (km == 0) hipLaunchKernelGGL((item_expand_fused_kernel<64, 0>), grid, dim3(256), 0, s, p);
    else if (km == 1) hipLaunchKernelGGL((item_expand_fused_kernel<64, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((item_expand_fused_kernel<64, 2>), grid, dim3(256), 0, s, p);
  } else if (D == 128) {
    if (km == 0) hipLaunchKernelGGL((item_expand_fused_kernel<128, 0>), grid, dim3(256), 0, s, p);
    else if (km == 1) hipLaunchKernelGGL((item_expand_fused_kernel<128, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((item_expand_fused_kernel<128, 2>), grid, dim3(256), 0, s, p);
  } else {
    if (km == 0) hipLaunchKernelGGL((item_expand_fused_kernel<256, 0>), grid, dim3(256), 0, s, p);
    else if (km == 1) hipLaunchKernelGGL((item_expand_fused_kernel<256, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((item_expand_fused_kernel<256, 2>), grid, dim3(256), 0, s, p);
  }
  DMM_LAUNCH_CHECK("dmm_item_expand_fused");
  return 0;
}
