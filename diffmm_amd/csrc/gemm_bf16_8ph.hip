// 256 x 256 x 64 bf16 NT GEMM tile for the reverse diffusion chain (same contract and epilogue as gemm_bf16.hip), built around
// LDS-DMA staging that stays in flight across barriers.  The register-staged tiles of gemm_bf16.hip top out where every K step
// drains its loads in front of a workgroup barrier; here
//   * 8 waves (2 along M x 4 along N, 128 x 64 outputs each = 32 MFMA 16x16x32 accumulators), one workgroup per CU, two waves per SIMD;
//   * the two wave rows run ONE BARRIER APART: while the waves of row 0 issue the 16 MFMAs of a phase, the waves of row 1 (one per
//     SIMD each) read their next fragments from LDS and issue the next LDS-DMA, and vice versa -- the matrix pipe of every SIMD
//     always has a wave in its MFMA cluster;
//   * a K tile is 4 phases (one 64 x 32 quadrant of the wave's outputs x K = 64 each); every phase reads at most one A unit and
//     one B unit and refills exactly one 16 KiB unit of LDS with data that is needed 5-6 phases later;
//   * completion is counted (s_waitcnt vmcnt(8): the four youngest units may still be in flight), never drained, and the
//     barriers are raw s_barrier -- a __syncthreads() would wait for every outstanding LDS-DMA.
//
// LDS (128 KiB): 2 K-tile buffers x 4 units x 16 KiB.  A unit is 128 rows x 64 bf16 (128-byte rows, 16-byte chunk c of row r at
// chunk c ^ ((r >> 1) & 7)); units are cut by WHEN they are read, not by where they sit in the tile:
//   X0 / X1: rows {qa*64 .. qa*64+63} of both wave rows' 128-row halves of the A tile   (read in phase 0 / phase 2)
//   Y0 / Y1: rows {qb*32 .. qb*32+31} of all four wave columns' 64-row quarters of B    (read in phase 0 / phase 1)
// so a unit is dead right after the single phase that reads it and can be refilled two phases later.
//
// Per K tile t (buffer t & 1); issue order is one unit per phase, forever: ... Y1(t+1), X1(t+1), X0(t+2), Y0(t+2), Y1(t+2) ...
//   phase 0: read Y0(t), X0(t)   issue Y1(t+1)   wait vmcnt(8)  -> Y1(t) has landed (needed in phase 1)     MFMA quadrant (0,0)
//   phase 1: read Y1(t)          issue X1(t+1)   wait vmcnt(8)  -> X1(t) has landed (needed in phase 2)     MFMA quadrant (0,1)
//   phase 2: read X1(t)          issue X0(t+2)   (nothing new is needed in phase 3)                          MFMA quadrant (1,1)
//   phase 3:                     issue Y0(t+2)   wait vmcnt(8)  -> X0(t+1), Y0(t+1) have landed              MFMA quadrant (1,0)
// Ordering rules (both follow from the wave rows being one barrier apart):
//   RAW: a wave's wait for its part of a unit sits in front of the FIRST barrier of the phase before the one that reads it;
//   WAR: a unit read in phase p is refilled in phase p+2 or later (the other row's reads of phase p retire one barrier later).
// Loads past the last K tile are still issued (clamped to the last tile, into units nobody reads any more) so the counts stay
// constant; everything is drained before the epilogue reuses LDS.
#include <string.h>

#include <algorithm>
#include <atomic>

#include "gemm_bf16.h"

namespace dmm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int P8_UNIT = 128 * 128;  // bytes
constexpr int P8_LDS = 8 * P8_UNIT;

#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P8_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

struct P8Acc {
  f32x4 v[2][2][4][2];  // [qa][qb][m tile][n tile]
};

template <int QA, int QB>
__device__ __forceinline__ void p8_mma(P8Acc& acc, const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc.v[QA][QB][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][ks], b[nt][ks], acc.v[QA][QB][mt][nt], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void p8_read_a(bf16x8 (&a)[4][2], const char* unit, int wr, const int (&lo)[2]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[mt][ks] = *reinterpret_cast<const bf16x8*>(unit + (wr * 64 + mt * 16) * 128 + lo[ks]);
}
__device__ __forceinline__ void p8_read_b(bf16x8 (&b)[2][2], const char* unit, int wc, const int (&lo)[2]) {
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b[nt][ks] = *reinterpret_cast<const bf16x8*>(unit + (wc * 32 + nt * 16) * 128 + lo[ks]);
}

// ---- epilogue ---------------------------------------------------------------------------------------------------------------
// 4 passes of 32 rows; every wave transposes its accumulators through its own LDS slab (row stride 68 floats: conflict-free for
// the accumulator-layout writes, 16-byte aligned for the row reads), then each lane owns 4 consecutive columns of 8 rows per pass.
// All operands of the epilogue are copied out of the kernel-argument struct ONCE (P8Epi, held in SGPRs): read through the struct
// reference the compiler re-fetched every field at every use (scalar load + full wait, ~300 of them: 27-44 us per workgroup,
// more than the K = 1024 main loop).
struct P8Epi {
  float* C; __bf16* Cb; const float* R; const __bf16* Rb; const float* bias; const float* bias2;
  long long ldc, ldcb, ldr, ldrb;
  int M, N, act;
  float alpha, e1, e2;
};
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int P8_LDT = 68;

template <int QA, int MP>
__device__ __forceinline__ void p8_acc_to_lds(const P8Acc& acc, float* tile, int q, int fr) {
#pragma unroll
  for (int ml = 0; ml < 2; ++ml)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(ml * 16 + q * 4 + r) * P8_LDT + qb * 32 + nt * 16 + fr] = acc.v[QA][qb][MP * 2 + ml][nt][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_wave_barrier();
}

// The epilogue's streams (fp32 state in and out, bf16 copy out: 640 KiB per tile, each byte touched once per launch) are marked
// non-temporal so they do not evict the A / B panels the other CUs of the XCD are re-reading from L2.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 p8_ld_stream(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void p8_st_stream(float* p, const float (&v)[4]) {
  __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void p8_st_stream(__bf16* p, const float (&v)[4]) {
  const bf16x4 b = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  __builtin_nontemporal_store(__builtin_bit_cast(u32x2, b), reinterpret_cast<u32x2*>(p));
}

// Interior tiles with 16-byte aligned operands: no bounds checks, running row pointers, and the residual rows of pass p+1 are
// requested before pass p stores (loads and stores retire in order on one counter, so a load issued behind a store waits for it).
// EDGE = false: interior tiles, no bounds checks.  EDGE = true: the same pipeline for tiles that reach past M or N -- a row predicate and,
// per lane, the number of valid columns among its four (`nv`, constant over the passes); partial lanes fall back to scalar accesses.
__device__ __forceinline__ float4 p8_ld_edge(const float* p, int nv) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nv > 0) r.x = p[0];
  if (nv > 1) r.y = p[1];
  if (nv > 2) r.z = p[2];
  return r;
}
template <typename T>
__device__ __forceinline__ void p8_st_edge(T* p, const float (&v)[4], int nv) {
  if (nv > 0) p[0] = (T)v[0];
  if (nv > 1) p[1] = (T)v[1];
  if (nv > 2) p[2] = (T)v[2];
}

// Residual kinds: RK = 0 none, 1 fp32 rows (R), 2 bf16 rows (Rb: the reverse chain with its running x_t kept in bf16 only).  pR is a byte
// pointer to the lane's four columns of the pass's first row, rs the row stride in bytes.
// A residual piece (the lane's four columns of one row) as it waits in registers: fp32 rows as a float4, bf16 rows as the raw 8 bytes
// (two passes of 8 pieces each are in flight per wave: 48 instead of 96 registers next to the 128 accumulators).
template <int RK> struct P8Res { typedef float4 type; };
template <> struct P8Res<2> { typedef u32x2 type; };
__device__ __forceinline__ float4 p8_res_f4(const float4& r) { return r; }
__device__ __forceinline__ float4 p8_res_f4(const u32x2& w) {
  return make_float4(__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u));
}
__device__ __forceinline__ void p8_res_zero(float4& r) { r = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void p8_res_zero(u32x2& r) { r = u32x2{0u, 0u}; }
__device__ __forceinline__ void p8_ld_res(const char* p, float4& r) { r = p8_ld_stream(reinterpret_cast<const float*>(p)); }
__device__ __forceinline__ void p8_ld_res(const char* p, u32x2& r) { r = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); }
__device__ __forceinline__ void p8_ld_res_edge(const char* p, int nv, float4& r) { r = p8_ld_edge(reinterpret_cast<const float*>(p), nv); }
__device__ __forceinline__ void p8_ld_res_edge(const char* p, int nv, u32x2& r) {
  const unsigned short* b = reinterpret_cast<const unsigned short*>(p);
  unsigned w0 = 0u, w1 = 0u;
  if (nv > 0) w0 = b[0];
  if (nv > 1) w0 |= (unsigned)b[1] << 16;
  if (nv > 2) w1 = b[2];
  r = u32x2{w0, w1};
}

// B2 (the latent reverse chain): the bf16 output is not a copy of C but the NEXT layer's activation tanh(C + bias2[n]).
// the persistent chain's h' stores.  KEEP = 1: plain stores (the lines stay in the XCD's L2 for the group's re-read; published by an agent
// release fence = L2 write-back).  KEEP = 2: write-through (sc1) stores, published by waiting for them (no release fence; the lines
// leave L2, the group re-reads them from the fabric side).  MI355X_MICROARCH.md "publish-large" prices both; measured here: see DESIGN 4.2.
template <int KEEP>  // 1: plain, 2: write-through
__device__ __forceinline__ void p8_st_keep(__bf16* p, const float (&v)[4]) {
  const bf16x4 b = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  if (KEEP == 2) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *reinterpret_cast<u32x2*>(p) = __builtin_bit_cast(u32x2, b);
}
template <int PASS, int RK, bool HAS_C, bool HAS_CB, int ACT, bool EDGE, bool B2, int KEEP = 0>
__device__ __forceinline__ void p8_fast_pass(const P8Epi& E, const P8Acc& acc, float* tile, int q, int fr, const float4& b4, const float4& c4, const char*& pR, long long rs,
                                             float*& pC, __bf16*& pCb, const typename P8Res<RK>::type (&rcur)[8], typename P8Res<RK>::type (&rnext)[8],
                                             int mrow, int nv) {
  constexpr bool HAS_R = RK != 0;
  if (HAS_R && PASS < 2) {  // two passes ahead: 16 KiB of residual rows in flight per wave while a pass is being stored
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const char* p = pR + (long long)(64 + it * 4) * rs;
      if (!EDGE) p8_ld_res(p, rnext[it]);
      else if (mrow + PASS * 32 + 64 + it * 4 < E.M) { if (nv == 4) p8_ld_res(p, rnext[it]); else p8_ld_res_edge(p, nv, rnext[it]); }
      else p8_res_zero(rnext[it]);
    }
  }
  p8_acc_to_lds<(PASS >> 1), (PASS & 1)>(acc, tile, q, fr);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const float4 a4 = *reinterpret_cast<const float4*>(tile + (it * 4 + q) * P8_LDT + fr * 4);
    float v[4] = {E.alpha * a4.x + b4.x, E.alpha * a4.y + b4.y, E.alpha * a4.z + b4.z, E.alpha * a4.w + b4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (ACT == DMM_ACT_TANH) v[u] = HAS_C ? tanhf(v[u]) : tanh_fast_bf16(v[u]);
      else if (ACT == DMM_ACT_SIGMOID) v[u] = 1.0f / (1.0f + expf(-v[u]));
    }
    if (HAS_R) {
      const float4 r4 = p8_res_f4(rcur[it]);
      v[0] = E.e1 * v[0] + E.e2 * r4.x; v[1] = E.e1 * v[1] + E.e2 * r4.y; v[2] = E.e1 * v[2] + E.e2 * r4.z; v[3] = E.e1 * v[3] + E.e2 * r4.w;
    }
    float w[4] = {v[0], v[1], v[2], v[3]};
    if (B2) { w[0] = tanh_fast_bf16(v[0] + c4.x); w[1] = tanh_fast_bf16(v[1] + c4.y); w[2] = tanh_fast_bf16(v[2] + c4.z); w[3] = tanh_fast_bf16(v[3] + c4.w); }
    if (!EDGE) {
      if (HAS_C) p8_st_stream(pC + (long long)(it * 4) * E.ldc, v);
      if (HAS_CB) { if (KEEP) p8_st_keep<KEEP>(pCb + (long long)(it * 4) * E.ldcb, w); else p8_st_stream(pCb + (long long)(it * 4) * E.ldcb, w); }
    } else if (mrow + PASS * 32 + it * 4 < E.M && nv > 0) {
      if (nv == 4) {
        if (HAS_C) p8_st_stream(pC + (long long)(it * 4) * E.ldc, v);
        if (HAS_CB) { if (KEEP) p8_st_keep<KEEP>(pCb + (long long)(it * 4) * E.ldcb, w); else p8_st_stream(pCb + (long long)(it * 4) * E.ldcb, w); }
      } else {
        if (HAS_C) p8_st_edge(pC + (long long)(it * 4) * E.ldc, v, nv);
        if (HAS_CB) p8_st_edge(pCb + (long long)(it * 4) * E.ldcb, w, nv);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // the pass's LDS reads are done before the next pass overwrites the slab
  __builtin_amdgcn_wave_barrier();
  if (HAS_R) pR += 32 * rs;
  if (HAS_C) pC += 32 * E.ldc;
  if (HAS_CB) pCb += 32 * E.ldcb;
}

template <int RK, bool HAS_C, bool HAS_CB, int ACT, bool EDGE, bool B2 = false, int KEEP = 0>
__device__ __forceinline__ void p8_epilogue_fast(const P8Epi& E, const P8Acc& acc, char* smem, int m0, int n0, int wave, int lane) {
  constexpr bool HAS_R = RK != 0;
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, q = lane >> 4;
  float* tile = reinterpret_cast<float*>(smem) + wave * 32 * P8_LDT;
  const int n = n0 + wc * 64 + fr * 4;
  const int mrow = m0 + wr * 128 + q;
  const long long m = mrow;
  const int nv = EDGE ? max(0, min(4, E.N - n)) : 4;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (E.bias) b4 = nv == 4 ? *reinterpret_cast<const float4*>(E.bias + n) : p8_ld_edge(E.bias + n, nv);
  float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (B2) c4 = nv == 4 ? *reinterpret_cast<const float4*>(E.bias2 + n) : p8_ld_edge(E.bias2 + n, nv);
  const long long rs = RK == 1 ? E.ldr * 4 : E.ldrb * 2;
  const char* pR = RK == 1 ? reinterpret_cast<const char*>(E.R + m * E.ldr + n) : RK == 2 ? reinterpret_cast<const char*>(E.Rb + m * E.ldrb + n) : nullptr;
  float* pC = HAS_C ? E.C + m * E.ldc + n : nullptr;
  __bf16* pCb = HAS_CB ? E.Cb + m * E.ldcb + n : nullptr;
  typename P8Res<RK>::type ra[8], rb[8], rc[8];
  if (HAS_R) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const char* p0 = pR + (long long)(it * 4) * rs;
      const char* p1 = pR + (long long)(32 + it * 4) * rs;
      if (!EDGE) {
        p8_ld_res(p0, ra[it]);
        p8_ld_res(p1, rb[it]);
      } else {
        if (mrow + it * 4 < E.M) { if (nv == 4) p8_ld_res(p0, ra[it]); else p8_ld_res_edge(p0, nv, ra[it]); } else p8_res_zero(ra[it]);
        if (mrow + 32 + it * 4 < E.M) { if (nv == 4) p8_ld_res(p1, rb[it]); else p8_ld_res_edge(p1, nv, rb[it]); } else p8_res_zero(rb[it]);
      }
    }
  }
  p8_fast_pass<0, RK, HAS_C, HAS_CB, ACT, EDGE, B2, KEEP>(E, acc, tile, q, fr, b4, c4, pR, rs, pC, pCb, ra, rc, mrow, nv);  // uses ra, requests pass 2 into rc
  p8_fast_pass<1, RK, HAS_C, HAS_CB, ACT, EDGE, B2, KEEP>(E, acc, tile, q, fr, b4, c4, pR, rs, pC, pCb, rb, ra, mrow, nv);  // uses rb, requests pass 3 into ra
  p8_fast_pass<2, RK, HAS_C, HAS_CB, ACT, EDGE, B2, KEEP>(E, acc, tile, q, fr, b4, c4, pR, rs, pC, pCb, rc, rb, mrow, nv);
  p8_fast_pass<3, RK, HAS_C, HAS_CB, ACT, EDGE, B2, KEEP>(E, acc, tile, q, fr, b4, c4, pR, rs, pC, pCb, ra, rb, mrow, nv);
}

// Edge tiles, unaligned operands and the rarely used operand combinations: bounds-checked, element-wise where needed.
template <int QA, int MP>
__device__ __forceinline__ void p8_generic_pass(const P8Epi& E, const P8Acc& acc, float* tile, int m0, int n0, int wave, int lane, bool vec_ok) {
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, q = lane >> 4;
  const int mbase = m0 + wr * 128 + QA * 64 + MP * 32, nbase = n0 + wc * 64;
  p8_acc_to_lds<QA, MP>(acc, tile, q, fr);
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + q;
    const int m = mbase + row, n = nbase + fr * 4;
    if (m >= E.M || n >= E.N) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(tile + row * P8_LDT + fr * 4);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const int nv = min(4, E.N - n);
    for (int u = 0; u < nv; ++u) {
      float x = E.alpha * v[u] + (E.bias ? E.bias[n + u] : 0.f);
      if (E.act == DMM_ACT_TANH) x = tanhf(x);
      else if (E.act == DMM_ACT_SIGMOID) x = 1.0f / (1.0f + expf(-x));
      if (E.R) x = E.e1 * x + E.e2 * E.R[(long long)m * E.ldr + n + u];
      else if (E.Rb) x = E.e1 * x + E.e2 * (float)E.Rb[(long long)m * E.ldrb + n + u];
      v[u] = x;
    }
    float w[4] = {v[0], v[1], v[2], v[3]};
    if (E.bias2) for (int u = 0; u < nv; ++u) w[u] = tanh_fast_bf16(v[u] + E.bias2[n + u]);
    if (nv == 4 && vec_ok) {
      if (E.C) *reinterpret_cast<float4*>(E.C + (long long)m * E.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      if (E.Cb) *reinterpret_cast<bf16x4*>(E.Cb + (long long)m * E.ldcb + n) = bf16x4{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
    } else {
      for (int u = 0; u < nv; ++u) {
        if (E.C) E.C[(long long)m * E.ldc + n + u] = v[u];
        if (E.Cb) E.Cb[(long long)m * E.ldcb + n + u] = (__bf16)w[u];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// Epilogue kinds.  Each gets its own kernel instantiation -- its own symbol in a rocprof trace (the chain's layer 1 and layer 2 no longer
// share one row of the kernel stats) and only its own epilogue code and registers.  EPI_GENERIC dispatches at run time.
enum { EPI_GENERIC = 0, EPI_TANH_CB = 1 /* chain layer 1 */, EPI_R32_C_CB = 2 /* chain layer 2, fp32 state */, EPI_R16_CB = 3 /* chain layer 2, bf16 state */,
       EPI_R16_C = 4 /* last chain step */, EPI_TANH_C_CB = 5 /* training: h in fp32 and bf16 */, EPI_C = 6, EPI_CB = 7, EPI_C_CB = 8 /* first step of a CSR-started chain */,
       EPI_LATENT = 9 /* latent chain step: z' = e1 (acc + g) + e2 z in fp32, tanh(z' + bias2) in bf16 */, EPI_LATENT_START = 10 /* its dense start: no residual */, EPI_COUNT = 11 };

static int p8_pick_epi(const GemmBf16P& P) {
  const bool noR = !P.R && !P.Rb;
  if (P.bias2) return P.act != DMM_ACT_NONE || P.Rb ? EPI_GENERIC : P.R ? EPI_LATENT : EPI_LATENT_START;
  if (noR && !P.C && P.Cb && P.act == DMM_ACT_TANH) return EPI_TANH_CB;
  if (P.R && !P.Rb && P.C && P.Cb && P.act == DMM_ACT_NONE) return EPI_R32_C_CB;
  if (P.Rb && !P.R && !P.C && P.Cb && P.act == DMM_ACT_NONE) return EPI_R16_CB;
  if (P.Rb && !P.R && P.C && !P.Cb && P.act == DMM_ACT_NONE) return EPI_R16_C;
  if (noR && P.C && P.Cb && P.act == DMM_ACT_TANH) return EPI_TANH_C_CB;
  if (noR && P.C && !P.Cb && P.act == DMM_ACT_NONE) return EPI_C;
  if (noR && !P.C && P.Cb && P.act == DMM_ACT_NONE) return EPI_CB;
  if (noR && P.C && P.Cb && P.act == DMM_ACT_NONE) return EPI_C_CB;
  return EPI_GENERIC;
}

template <int EPI>
__device__ __forceinline__ void p8_epilogue(const GemmBf16P& P, const P8Acc& acc, char* smem, int m0, int n0, int wave, int lane) {
  P8Epi E;
  E.C = P.C; E.Cb = P.Cb; E.R = P.R; E.Rb = P.Rb; E.bias = P.bias; E.bias2 = P.bias2;
  E.ldc = P.ldc; E.ldcb = P.ldcb; E.ldr = P.ldr; E.ldrb = P.ldrb;
  E.M = P.M; E.N = P.N; E.act = P.act; E.alpha = P.alpha; E.e1 = P.e1; E.e2 = P.e2;
  const bool vec_ok = (!E.C || (E.ldc % 4 == 0 && ((uintptr_t)E.C & 15) == 0)) && (!E.Cb || (E.ldcb % 4 == 0 && ((uintptr_t)E.Cb & 7) == 0));
  const bool fast = vec_ok && (!E.bias || ((uintptr_t)E.bias & 15) == 0) && (!E.bias2 || ((uintptr_t)E.bias2 & 15) == 0) && (!E.R || (E.ldr % 4 == 0 && ((uintptr_t)E.R & 15) == 0)) &&
                    (!E.Rb || (!E.R && E.ldrb % 4 == 0 && ((uintptr_t)E.Rb & 7) == 0));
  const bool interior = m0 + 256 <= E.M && n0 + 256 <= E.N;
#define P8_FAST(R_, C_, CB_, ACT_)                                                                   \
  do {                                                                                               \
    if (interior) p8_epilogue_fast<R_, C_, CB_, ACT_, false>(E, acc, smem, m0, n0, wave, lane);      \
    else p8_epilogue_fast<R_, C_, CB_, ACT_, true>(E, acc, smem, m0, n0, wave, lane);                \
  } while (0)
  if (fast && EPI == EPI_TANH_CB) {
    P8_FAST(0, false, true, DMM_ACT_TANH);
  } else if (fast && EPI == EPI_CB) {
    P8_FAST(0, false, true, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_R32_C_CB) {
    P8_FAST(1, true, true, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_R16_CB) {  // reverse chain, bf16 running state: x_t in, x_{t-1} out, both bf16
    P8_FAST(2, false, true, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_R16_C) {  // its last step: the fp32 view the top-k reads
    P8_FAST(2, true, false, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_C) {
    P8_FAST(0, true, false, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_C_CB) {
    P8_FAST(0, true, true, DMM_ACT_NONE);
  } else if (fast && EPI == EPI_LATENT) {
    if (interior) p8_epilogue_fast<1, true, true, DMM_ACT_NONE, false, true>(E, acc, smem, m0, n0, wave, lane);
    else p8_epilogue_fast<1, true, true, DMM_ACT_NONE, true, true>(E, acc, smem, m0, n0, wave, lane);
  } else if (fast && EPI == EPI_LATENT_START) {
    if (interior) p8_epilogue_fast<0, true, true, DMM_ACT_NONE, false, true>(E, acc, smem, m0, n0, wave, lane);
    else p8_epilogue_fast<0, true, true, DMM_ACT_NONE, true, true>(E, acc, smem, m0, n0, wave, lane);
  } else if (fast && EPI == EPI_TANH_C_CB) {  // training forward: h in fp32 (for the backward) and bf16 (next operand)
    P8_FAST(0, true, true, DMM_ACT_TANH);
#undef P8_FAST
  } else {  // unaligned operands and the rarely used operand combinations
    float* tile = reinterpret_cast<float*>(smem) + wave * 32 * P8_LDT;
    p8_generic_pass<0, 0>(E, acc, tile, m0, n0, wave, lane, vec_ok);
    p8_generic_pass<0, 1>(E, acc, tile, m0, n0, wave, lane, vec_ok);
    p8_generic_pass<1, 0>(E, acc, tile, m0, n0, wave, lane, vec_ok);
    p8_generic_pass<1, 1>(E, acc, tile, m0, n0, wave, lane, vec_ok);
  }
}

#ifdef P8_TIMING
__device__ unsigned long long p8_stamps[8 * 8192];
#define P8_STAMP(i) if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 8192) p8_stamps[(blockIdx.x * 8 + (i))] = wall_clock64()
// the persistent chain kernel: 8 stamps per (workgroup, step), workgroups < 512, steps < 64 (tools/latent_stamps.py)
__device__ unsigned long long latent_stamps[512 * 64 * 8];
#define LAT_STAMP(step, i) if (threadIdx.x == 0 && blockIdx.x < 512 && (step) < 64) latent_stamps[((blockIdx.x * 64 + (step)) * 8 + (i))] = wall_clock64()
#else
#define P8_STAMP(i)
#define LAT_STAMP(step, i)
#endif

// The K loop of one 256 x 256 tile: LDS-DMA prologue, nk K tiles of 4 phases, drain.  On return every LDS-DMA has landed and every fragment
// read is done (LDS is free) and the wave rows are re-aligned.  acc is accumulated into (the caller initialises it).
template <int STAMP>  // 0: none, 1: the GEMM kernel's stamp 1, 2: the chain kernel's stamp 1 of step `stamp_step` (both P8_TIMING builds only)
__device__ __forceinline__ void p8_main_loop(P8Acc& acc, const __bf16* const (&offX)[2][2], const __bf16* const (&offY)[2][2], int nk, char* smem, int wave,
                                             int wr, int wc, const int (&lo)[2], [[maybe_unused]] int stamp_step = 0) {
  [[maybe_unused]] const int lane = threadIdx.x & 63;  // (P8_STAMP)
  auto issue = [&](const __bf16* const (&src)[2], int kt, char* unit) {
    const int k0 = (kt < nk ? kt : nk - 1) * 64;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[0] + k0), (lds_ptr_t)(unit + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[1] + k0), (lds_ptr_t)(unit + (wave + 8) * 1024), 16, 0, 0);
  };
  char* const buf0 = smem;
  char* const buf1 = smem + 4 * P8_UNIT;
  constexpr int X0 = 0, X1 = P8_UNIT, Y0 = 2 * P8_UNIT, Y1 = 3 * P8_UNIT;
  // prologue: the issue sequence up to (not including) phase 0 of tile 0
  issue(offX[0], 0, buf0 + X0);
  issue(offY[0], 0, buf0 + Y0);
  issue(offY[1], 0, buf0 + Y1);
  issue(offX[1], 0, buf0 + X1);
  issue(offX[0], 1, buf1 + X0);
  issue(offY[0], 1, buf1 + Y0);
  P8_VMCNT8();   // X0(0), Y0(0) have landed (this wave's part)
  P8_BARRIER();  // ... every wave's part
  if (STAMP == 1 && threadIdx.x < 64) { P8_STAMP(1); }
  if (STAMP == 2) { LAT_STAMP(stamp_step, 1); }
  if (wr == 1) P8_BARRIER();  // row 1 runs one barrier behind row 0 from here on

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  for (int t = 0; t < nk; ++t) {
    char* const cur = (t & 1) ? buf1 : buf0;
    char* const oth = (t & 1) ? buf0 : buf1;
    // ---- phase 0
    p8_read_b(fb0, cur + Y0, wc, lo);
    p8_read_a(fa, cur + X0, wr, lo);
    issue(offY[1], t + 1, oth + Y1);
    P8_VMCNT8();
    P8_BARRIER();
    P8_LGKM0();
    p8_mma<0, 0>(acc, fa, fb0);
    P8_BARRIER();
    // ---- phase 1
    p8_read_b(fb1, cur + Y1, wc, lo);
    issue(offX[1], t + 1, oth + X1);
    P8_VMCNT8();
    P8_BARRIER();
    P8_LGKM0();
    p8_mma<0, 1>(acc, fa, fb1);
    P8_BARRIER();
    // ---- phase 2
    p8_read_a(fa, cur + X1, wr, lo);
    issue(offX[0], t + 2, cur + X0);
    P8_BARRIER();
    P8_LGKM0();
    p8_mma<1, 1>(acc, fa, fb1);
    P8_BARRIER();
    // ---- phase 3
    issue(offY[0], t + 2, cur + Y0);
    P8_VMCNT8();
    P8_BARRIER();
    p8_mma<1, 0>(acc, fa, fb0);
    P8_BARRIER();
  }
  if (wr == 0) P8_BARRIER();  // row 0 waits for row 1's last phase
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every LDS-DMA landed, every read done: LDS is free
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_8ph_kernel(const GemmBf16Batch batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmBf16P& P = batch.p[blockIdx.y];
  // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each) and an XCD runs ~32 of them at a time.
  // xcd_remap gives every XCD one contiguous run of logical tile ids; the ids walk the tile grid in SUPERTILES of SM x SN tiles
  // (SM * SN ~ 32), so the tiles an XCD has in flight share SM row panels of A and SN row panels of B instead of streaming all
  // of B for every row panel (row-major order: (1 + 32) panels per 32 tiles; 4 x 8 supertiles: 12).
  int mtile, ntile;
  {
    const int bid = P.part ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    if (P.part == 2) {
      const int SN = P.nt < 8 ? P.nt : 8, SM = 32 / SN > 0 ? 32 / SN : 1;
      const int band = bid / (SM * P.nt);
      const int hb = min(SM, P.mt - band * SM);
      const int i = bid - band * SM * P.nt;
      const int c = i / (hb * SN), r = i - c * hb * SN;
      const int w = min(SN, P.nt - c * SN);
      mtile = band * SM + r / w;
      ntile = c * SN + r % w;
    } else {
      mtile = bid / P.nt;
      ntile = bid % P.nt;
    }
  }
  const int m0 = mtile * 256, n0 = ntile * 256;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, q = lane >> 4;
  const int nk = P.K / 64;
  if (threadIdx.x < 64) { P8_STAMP(0); }

  // ---- staging addresses: wave w fills row groups w and w + 8 (8 rows = 1 KiB each) of every unit
  // (full per-lane pointers, held in registers: the asm barriers clobber memory, so anything read through P would be re-fetched)
  const __bf16* offX[2][2];  // [unit][load]
  const __bf16* offY[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = 8 * (wave + 8 * i) + (lane >> 3);       // unit row this lane fills
    const int cs = ((lane & 7) ^ ((u >> 1) & 7)) * 8;     // source chunk (elements) that belongs at LDS chunk (lane & 7)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
      const int rb = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
      offX[h][i] = P.A + (long long)(ra < P.M ? ra : P.M - 1) * P.lda + cs;
      offY[h][i] = P.B + (long long)(rb < P.N ? rb : P.N - 1) * P.ldb + cs;
    }
  }
  // ---- fragment read offsets inside a unit (the row-dependent swizzle term depends only on the lane: rows step by 16)
  int lo[2];
  {
    const int sw = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) lo[ks] = fr * 128 + (((q | (ks << 2)) ^ sw) << 4);
  }

  P8Acc acc;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc.v[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  p8_main_loop<1>(acc, offX, offY, nk, smem, wave, wr, wc, lo);
  if (threadIdx.x < 64) { P8_STAMP(2); }
  p8_epilogue<EPI>(P, acc, smem, m0, n0, wave, lane);
  if (threadIdx.x < 64) { P8_STAMP(3); }
#ifdef P8_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x < 64) { P8_STAMP(4); }
#endif
}

// ---- latent reverse chain: all T - 1 steps of  z' = c2 z + c1 (h G^T + g),  h' = tanh(z' + bias_t)  in ONE launch -------------------------------
// A step launched on its own (EPI_LATENT) is bound by streaming the fp32 state z through HBM: 8 of its 12 bytes per element.  Here a workgroup keeps its
// 256 x 256 tile of z in the accumulator registers for the whole chain (as a = z / c1: the MFMAs of a step accumulate h G^T straight onto
// a_init = (c2 / c1) z + g, and the next step's a_init is one FMA per register) -- z never touches memory again.  What a step still exchanges is h: the
// H / 256 workgroups that share a row block (a "group") each write their 256 columns of h' (bf16) and read the whole row block back as the next A
// operand, so a step ends in a group barrier: the stores are published (WT: write-through sc1 stores that every wave waits for; else plain stores +
// one agent release fence), one atomic per workgroup on the group's counter, a BOUNDED spin (a partner that never arrives -- it cannot, see below --
// sets the error word instead of hanging the GPU), acquire fence.
// Placement: workgroup b belongs to group 8 (b / 8 / nt) + b % 8 -- the members of a group are nt CONSECUTIVE workgroups of one XCD's dispatch queue
// (workgroups are dealt round-robin over the 8 XCDs), so they share that XCD's L2 (h' is re-read from it) and, dispatch being in order per XCD, the
// earliest unfinished group of an XCD always has all its members resident: no wait can be circular.  The host launches at most CUs / nt groups at a time.
struct LatentP {
  const __bf16* G; long long ldg;   // [H, H]
  const float* g;                   // [H]
  const float* z; long long ldz;    // [rows, H] fp32: z_T, read once
  __bf16* ha; __bf16* hb; long long ldh;  // [rows, H] bf16 ping-pong: ha holds h_{T-1} on entry
  const float* bias_t; long long ldbt;    // [T, H]
};
struct LatentBatch {
  LatentP p[GB_MAX_BATCH];
  const float* coef;  // [2 T]: coef[2 i] = c1[i] (h' = tanh(c1[i] a + bias_t[i - 1])), coef[2 i + 1] = c2[i - 1] c1[i] / c1[i - 1] (a <- that * a + g)
  int* flags;         // [0, 256): group counters, [256, 512): per-group XCC-id masks (both zero on entry), flags[err_at] = error word
  float s_init;       // c2[T - 1] / c1[T - 1]
  int rows, H, T, rbs, g0, n_groups, err_at;
  int contig;                   // groups dealt to the XCDs in contiguous runs (1) or strided (0)
  int first_wave, phase_ticks;  // second form: workgroup slots per XCD that get a CU to themselves; start delay of the later ones (0: none)
};
constexpr int LATENT_SPIN_MAX = 1 << 22;  // x (s_sleep(2) + an L2 round trip, ~0.3 us): about a second, then give up (the step kernel path takes a few ms in total)

template <bool WT, bool EARLY_INV>
__global__ __launch_bounds__(512) void latent_chain_8ph_kernel(const LatentBatch Lb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = Lb.H >> 8;
  const int slot = blockIdx.x >> 3;
  // group of this workgroup: strided over the XCDs (consecutive groups on different XCDs), or -- Lb.contig -- each XCD takes a CONTIGUOUS run of
  // groups, i.e. (mostly) one modality: that modality's G (2 MiB at H = 1024) then stays in the XCD's 4 MiB L2 instead of two or three of them
  // competing for it with the h exchange
  const int gpx = (Lb.n_groups + 7) >> 3;
  const int gl = Lb.contig ? (int)(blockIdx.x & 7) * gpx + slot / nt : (slot / nt) * 8 + (int)(blockIdx.x & 7);
  if (gl >= Lb.n_groups || (Lb.contig && slot / nt >= gpx)) return;  // (uniform over the workgroup; such workgroups belong to no group)
  const int group = Lb.g0 + gl;
  const LatentP& P = Lb.p[group / Lb.rbs];
  const int m0 = (group % Lb.rbs) * 256, n0 = (slot % nt) * 256;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, q = lane >> 4;
  const int nk = Lb.H / 64, rows = Lb.rows, T = Lb.T;
  const float* const coef = Lb.coef;
  const float* const bias_t = P.bias_t;
  const long long ldbt = P.ldbt, ldh = P.ldh;
  __bf16* const hbuf[2] = {P.ha, P.hb};
  int* const flag = Lb.flags + gl;
  int* const err = Lb.flags + Lb.err_at;
  // placement check: the members of a group are expected on ONE XCD (they re-read each other's h' from its L2).  Every member ORs its
  // XCC id into the group's mask before its first arrival; after the first barrier a mask with more than one bit sets error word bit 1
  // (the hand-off below is agent-scope and stays correct across XCDs; the bit reports that the speed assumption did not hold).
  if (threadIdx.x == 0)
    __hip_atomic_fetch_or(Lb.flags + 256 + gl, 1 << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  const __bf16* offA[2][2];  // [unit][load] into h_a; h_b is `hdelta` elements further
  const __bf16* offY[2][2];
  const long long hdelta = P.hb - P.ha;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = 8 * (wave + 8 * i) + (lane >> 3);
    const int cs = ((lane & 7) ^ ((u >> 1) & 7)) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ra = m0 + (u >> 6) * 128 + h * 64 + (u & 63);
      const int rb = n0 + (u >> 5) * 64 + h * 32 + (u & 31);
      const long long ro = (long long)(ra < rows ? ra : rows - 1) * ldh + cs;
      offA[h][i] = P.ha + ro;
      offY[h][i] = P.G + (long long)rb * P.ldg + cs;
    }
  }
  int lo[2];
  {
    const int sw = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) lo[ks] = fr * 128 + (((q | (ks << 2)) ^ sw) << 4);
  }
  // a_init = (c2 / c1)[T - 1] z_T + g in the accumulator layout: acc.v[qa][qb][c][n][r] is row m0 + wr 128 + qa 64 + c 16 + q 4 + r,
  // column n0 + wc 64 + qb 32 + n 16 + fr (p8_acc_to_lds)
  P8Acc acc;
  float gcol[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int n = 0; n < 2; ++n) gcol[qb][n] = P.g[n0 + wc * 64 + qb * 32 + n * 16 + fr];
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 128 + qa * 64 + c * 16 + q * 4 + r;
        const float* zr = P.z + (long long)(row < rows ? row : rows - 1) * P.ldz + n0 + wc * 64 + fr;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc.v[qa][qb][c][n][r] = Lb.s_init * zr[qb * 32 + n * 16] + gcol[qb][n];
      }

  const bool interior = m0 + 256 <= rows;
  for (int step = 0; step < T - 1; ++step) {
    const int i = T - 1 - step, src = step & 1;
    {
      const long long hd = src ? hdelta : 0;
      const __bf16* offX[2][2] = {{offA[0][0] + hd, offA[0][1] + hd}, {offA[1][0] + hd, offA[1][1] + hd}};
      LAT_STAMP(step, 0);
      p8_main_loop<2>(acc, offX, offY, nk, smem, wave, wr, wc, lo, step);
      LAT_STAMP(step, 2);
    }
    // The acquire's L1 invalidate, HOISTED in front of the epilogue (DMM_LATENT_EARLY_INV, default on): this CU's loads of h are over for this step
    // (the main loop ended with vmcnt(0) + barrier) and none is issued before the group barrier below has been passed, so no line of either h buffer
    // can enter L1 between this invalidate and the next step's loads -- the stale copies the acquire exists to drop (lines of the buffer the next
    // step reads, cached when it was last read two steps ago) are dropped here, and the 1.5-1.7 us the invalidate takes (MI355X_MICROARCH.md,
    // inter-workgroup visibility) pass under the epilogue instead of behind the poll.  Its completion is waited for by wave 0's vmcnt(0) below.
    if (EARLY_INV && threadIdx.x == 0) asm volatile("buffer_inv sc1" ::: "memory");
    P8Epi E;
    E.C = nullptr; E.Cb = hbuf[src ^ 1]; E.R = nullptr; E.Rb = nullptr; E.bias = nullptr; E.bias2 = bias_t + (long long)(i - 1) * ldbt;
    E.ldc = 0; E.ldcb = ldh; E.ldr = 0; E.ldrb = 0; E.M = rows; E.N = Lb.H; E.act = DMM_ACT_NONE; E.alpha = coef[2 * i]; E.e1 = 1.f; E.e2 = 1.f;
    if (interior) p8_epilogue_fast<0, false, true, DMM_ACT_NONE, false, true, (WT ? 2 : 1)>(E, acc, smem, m0, n0, wave, lane);
    else p8_epilogue_fast<0, false, true, DMM_ACT_NONE, true, true, (WT ? 2 : 1)>(E, acc, smem, m0, n0, wave, lane);
    LAT_STAMP(step, 3);
    if (step + 1 == T - 1) break;  // h_0 is written; nothing follows in this kernel
    const float sc = coef[2 * i + 1];
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.v[qa][qb][c][n][r] = sc * acc.v[qa][qb][c][n][r] + gcol[qb][n];
    // ---- group barrier (MI355X_MICROARCH.md, inter-workgroup visibility): plain stores (h' stays in this XCD's L2 for the group's re-read) ->
    // every wave waits for its stores -> workgroup barrier -> ONE lane: agent release (L2 write-back), wait, arrive, poll, agent acquire (this
    // CU's L1 invalidated), wait -> workgroup barrier -> the next step's loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LAT_STAMP(step, 4);
    if (threadIdx.x == 0) {
      if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait: the flag must not overtake the write-back)
      __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (step + 1) * nt;
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > LATENT_SPIN_MAX) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      if (step == 0) {
        const int mask = __hip_atomic_load(Lb.flags + 256 + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mask & (mask - 1)) __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!EARLY_INV) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      LAT_STAMP(step, 5);
    }
    __syncthreads();
  }
}

// ---- the persistent chain, second form: TWO workgroups per CU ------------------------------------------------------------------------------------
// Per-workgroup timestamps of the kernel above (tools/latent_stamps.py) split a step of 32 us into 21 us of main loop and 11 us in which the matrix
// pipe idles: epilogue 4.7 (bound by the exp / rcp of the tanh: VALU, not memory), store drain 2.2, group barrier 2.4, prologue 1.  One workgroup per
// CU cannot overlap those phases with anything.  Here a workgroup is FOUR waves (one per SIMD) that keep a 128 x 256 tile of z (128 accumulator
// registers per lane, wave tile 64 x 128) and two workgroups share a CU: while one runs its epilogue / waits at its group barrier, the other one's
// MFMAs have the matrix pipe to themselves.  LDS per workgroup 72 KiB: a ring of three stages of K = 32 (A: 128 rows x 64 B, B: 256 rows x 64 B,
// linear images -- a 16 x 32 bf16 fragment read is 1 KiB contiguous, conflict-free without a swizzle), filled by LDS-DMA two stages ahead with counted
// completion (vmcnt(6): the next stage's six loads per wave may still be in flight) and ONE raw barrier per stage.  Registers <= 256 per lane.
// Group = the H / 256 workgroups of a 128-row block; placement, hand-off and error word exactly as above.  The host launches at most 2 CUs / nt groups
// at a time (every member workgroup resident: two per CU, checked by an occupancy query; otherwise the first form runs).
constexpr int V2_KT = 32;                           // K per stage
constexpr int V2_A_BYTES = 128 * V2_KT * 2;         // 8 KiB
constexpr int V2_STAGE = (128 + 256) * V2_KT * 2;   // 24 KiB
constexpr int V2_STAGES = 3;
constexpr int V2_LDS = V2_STAGES * V2_STAGE;        // 72 KiB
constexpr int V2_SLAB_LD = 132;                     // epilogue slab: 32 rows x 128 columns of fp32 per wave, row stride 132 floats (conflict-free writes)

#ifdef P8_TIMING
#define LAT2_STAMP(step, i) if (threadIdx.x == 0 && blockIdx.x < 512 && (step) < 64) latent_stamps[((blockIdx.x * 64 + (step)) * 8 + (i))] = wall_clock64()
// inside one stage (stage 8 of step 5), in shader clocks: slots (60, 0..5) of the workgroup's stamp rows
#define LAT2_CYC(i) if (threadIdx.x == 0 && blockIdx.x < 512 && step == 5 && t == 8) latent_stamps[((blockIdx.x * 64 + 60) * 8 + (i))] = __builtin_readcyclecounter()
#else
#define LAT2_STAMP(step, i)
#define LAT2_CYC(i)
#endif

// h' store of the second form: 4 bf16 = 8 bytes per lane.  Inline assembly for the same reason as its fragment reads: as a C++ (atomic) store through
// a pointer the compiler cannot prove global it becomes flat_store with s_waitcnt vmcnt(0) on both sides -- every store serialised (14 us of epilogue).
// Addressing: a wave-uniform base in SGPRs + a 32-bit per-lane byte offset (one register that walks the rows; 32 precomputed 64-bit row pointers
// were hoisted out of the step loop and spilled).
template <bool WT>
__device__ __forceinline__ void v2_store_h(const __bf16* sbase, unsigned voff, const float (&v)[4]) {
  const bf16x4 b = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  const u32x2 d = __builtin_bit_cast(u32x2, b);
  if (WT) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff), "v"(d), "s"(sbase) : "memory");  // write-through, agent scope
  else asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(d), "s"(sbase) : "memory");
}

template <bool WT, bool EDGE>
__device__ __forceinline__ void v2_epilogue(const f32x4 (&acc)[4][8], float* slab, float alpha, const float4& bb, const __bf16* sdst, unsigned voff, unsigned ldh,
                                            int rows_left, int q, int fr, int lane) {
  const unsigned row_step = 2u * ldh * 2u;  // two rows per store instruction
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
      for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(ml * 16 + q * 4 + r) * V2_SLAB_LD + n * 16 + fr] = acc[pass * 2 + ml][n][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // a lane owns 4 consecutive columns of 16 rows of the pass: 32 lanes cover a row's 128 columns (256 bytes of bf16: two whole 128-byte
    // lines per store instruction and row), two rows per instruction
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int rl = it * 2 + (lane >> 5);  // row of the pass
      const float4 a4 = *reinterpret_cast<const float4*>(slab + rl * V2_SLAB_LD + (lane & 31) * 4);
      const float w[4] = {tanh_fast_bf16(fmaf(alpha, a4.x, bb.x)), tanh_fast_bf16(fmaf(alpha, a4.y, bb.y)), tanh_fast_bf16(fmaf(alpha, a4.z, bb.z)),
                          tanh_fast_bf16(fmaf(alpha, a4.w, bb.w))};
      if (!EDGE || pass * 32 + rl < rows_left) v2_store_h<WT>(sdst, voff, w);
      voff += row_step;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the pass's LDS reads are done before the next pass overwrites the slab
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool WT>
__global__ __launch_bounds__(256, 2) void latent_chain_v2_kernel(const LatentBatch Lb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = Lb.H >> 8;
  const int slot = blockIdx.x >> 3;
  // group of this workgroup: strided over the XCDs (consecutive groups on different XCDs), or -- Lb.contig -- each XCD takes a CONTIGUOUS run of
  // groups, i.e. (mostly) one modality: that modality's G (2 MiB at H = 1024) then stays in the XCD's 4 MiB L2 instead of two or three of them
  // competing for it with the h exchange
  const int gpx = (Lb.n_groups + 7) >> 3;
  const int gl = Lb.contig ? (int)(blockIdx.x & 7) * gpx + slot / nt : (slot / nt) * 8 + (int)(blockIdx.x & 7);
  if (gl >= Lb.n_groups || (Lb.contig && slot / nt >= gpx)) return;  // (uniform over the workgroup; such workgroups belong to no group)
  const int group = Lb.g0 + gl;
  const LatentP& P = Lb.p[group / Lb.rbs];
  const int m0 = (group % Lb.rbs) * 128, n0 = (slot % nt) * 256;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, q = lane >> 4;
  const int nk = Lb.H / V2_KT, rows = Lb.rows, T = Lb.T;
  const float* const coef = Lb.coef;
  const float* const bias_t = P.bias_t;
  const long long ldbt = P.ldbt, ldh = P.ldh;
  __bf16* const hbuf[2] = {P.ha, P.hb};
  int* const flag = Lb.flags + gl;
  int* const err = Lb.flags + Lb.err_at;
  if (threadIdx.x == 0)
    __hip_atomic_fetch_or(Lb.flags + 256 + gl, 1 << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- staging: a wave-load (buffer_load ... lds) moves 16 rows x 64 B (lane -> row lane / 4, 16-byte chunk lane % 4) to 1 KiB of LDS; wave w
  // fills row blocks {w, w + 4} of A (8 blocks) and {w, w + 4, w + 8, w + 12} of B (16 blocks) of every stage.  Buffer addressing keeps the per-lane
  // part of an address in ONE register per operand (the row block and the K offset ride in the scalar offset) and returns zeros for rows past the
  // operand's end (the ragged last row block needs no clamp).
  const unsigned h_bytes = (unsigned)((unsigned long long)rows * (unsigned long long)ldh * 2ull);
  const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)P.ha, 0, h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)P.hb, 0, h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.G, 0, (unsigned)((unsigned long long)Lb.H * (unsigned long long)P.ldg * 2ull), 0x00020000);
  const unsigned voffA = (unsigned)(lane >> 2) * (unsigned)ldh * 2u + (unsigned)(lane & 3) * 16u;
  const unsigned voffB = (unsigned)(lane >> 2) * (unsigned)P.ldg * 2u + (unsigned)(lane & 3) * 16u;
  const unsigned sA = (unsigned)((unsigned long long)(m0 + wave * 16) * (unsigned long long)ldh * 2ull), sA_step = 64u * (unsigned)ldh * 2u;
  const unsigned sB = (unsigned)(n0 + wave * 16) * (unsigned)P.ldg * 2u, sB_step = 64u * (unsigned)P.ldg * 2u;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const int offA = (wr * 64 + fr) * 64 + q * 16;                   // fragment of m-tile mt: + mt * 1024
  const int offB = V2_A_BYTES + (wc * 128 + fr) * 64 + q * 16;     // fragment of n-tile nt: + nt * 1024

  // a_init = (c2 / c1)[T - 1] z_T + g in the accumulator layout: acc[mt][n][r] is row m0 + wr 64 + mt 16 + q 4 + r, column n0 + wc 128 + n 16 + fr
  f32x4 acc[4][8];
  const float* const gp = P.g + n0 + wc * 128 + fr;  // g of the lane's columns: gp[n * 16] (re-read per step: 8 registers less across the main loop)
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const float gn = gp[n * 16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + mt * 16 + q * 4 + r;
        acc[mt][n][r] = Lb.s_init * P.z[(long long)(row < rows ? row : rows - 1) * P.ldz + n0 + wc * 128 + fr + n * 16] + gn;
      }
  }

  float* const slab = reinterpret_cast<float*>(smem) + wave * (32 * V2_SLAB_LD);
  const bool interior = m0 + 128 <= rows;
  // De-phasing.  Two workgroups that share a CU start together and run the same program: left alone they stay in lockstep -- both in their main loops
  // (each at half the matrix rate), then both in their epilogues (no MFMA at all) -- and the second workgroup buys nothing.  The workgroups an XCD
  // receives after its first `Lb.first_wave` (one per CU) are the ones that land next to an earlier workgroup; they -- whole groups, slot / nt being
  // the group -- start `Lb.phase_ticks` (100 MHz ticks, about half a step) late, so that one workgroup's epilogue / barrier falls into the other's
  // main loop.  A group's own members stay in phase through their barrier; the two workgroups of a CU belong to different groups.
  if (Lb.phase_ticks > 0 && (slot / nt) * nt >= Lb.first_wave) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)Lb.phase_ticks) __builtin_amdgcn_s_sleep(32);
  }
  for (int step = 0; step < T - 1; ++step) {
    const int i = T - 1 - step, src = step & 1;
    LAT2_STAMP(step, 0);
    // ---- main loop: nk stages of K = 32
    auto issue = [&](int t, int buf) {
      const unsigned k0 = (unsigned)t * (V2_KT * 2);
      char* const st = smem + buf * V2_STAGE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (src) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (lds_ptr_t)(st + (wave + 4 * j) * 1024), 16, voffA, sA + j * sA_step + k0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, (lds_ptr_t)(st + (wave + 4 * j) * 1024), 16, voffA, sA + j * sA_step + k0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(st + V2_A_BYTES + (wave + 4 * j) * 1024), 16, voffB, sB + j * sB_step + k0, 0, 0);
    };
    issue(0, 0);
    issue(1, 1);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      LAT2_CYC(0);
      if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // this wave's part of stage t has landed (stage t + 1 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LAT2_CYC(1);
      P8_BARRIER();  // every wave's part of stage t; and every wave is done reading stage t - 1, whose buffer the next issue refills
      LAT2_CYC(2);
      if (t == 0) { LAT2_STAMP(step, 1); }
      if (t + 2 < nk) issue(t + 2, buf == 0 ? 2 : buf - 1);
      LAT2_CYC(3);
      // The fragment reads are INLINE ASSEMBLY: written as C++ loads, the compiler orders every LDS read behind ALL outstanding `buffer_load ... lds`
      // (it cannot tell the ring's stages apart and inserts s_waitcnt vmcnt(0) in front of the reads -- the loads issued two lines above included:
      // no prefetch at all; measured 20 us of main loop for half a tile).  The counted waits below are therefore explicit too.
      const unsigned aA = lds_base + (unsigned)buf * V2_STAGE + (unsigned)offA, aB = lds_base + (unsigned)buf * V2_STAGE + (unsigned)offB;
      bf16x8 fa[4], fb[8];
#define V2_DS(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
      V2_DS(fa[0], aA, 0); V2_DS(fa[1], aA, 1024); V2_DS(fa[2], aA, 2048); V2_DS(fa[3], aA, 3072);
      V2_DS(fb[0], aB, 0); V2_DS(fb[1], aB, 1024); V2_DS(fb[2], aB, 2048); V2_DS(fb[3], aB, 3072);
      V2_DS(fb[4], aB, 4096); V2_DS(fb[5], aB, 5120); V2_DS(fb[6], aB, 6144); V2_DS(fb[7], aB, 7168);
#undef V2_DS
      // (no s_setprio around the MFMAs: the wave that shares this SIMD belongs to the OTHER workgroup and may be in its epilogue -- MI355X_MICROARCH.md
      // "two waves per SIMD" item 2)
#define V2_MMA(n, cnt)                                                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")" ::: "memory");                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                                               \
      _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[n], acc[mt][n], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      LAT2_CYC(4);
      V2_MMA(0, 7) V2_MMA(1, 6) V2_MMA(2, 5) V2_MMA(3, 4) V2_MMA(4, 3) V2_MMA(5, 2) V2_MMA(6, 1) V2_MMA(7, 0)
#undef V2_MMA
      LAT2_CYC(5);
      buf = buf == 2 ? 0 : buf + 1;
    }
    P8_BARRIER();  // every wave is done reading the last stage: LDS is free for the epilogue slabs
    // (every LDS-DMA has landed -- the last stage waited for vmcnt(0) -- but only a wait the COMPILER can see clears its scoreboard: without this one
    // it orders each slab read of the epilogue behind the "outstanding" DMA with s_waitcnt vmcnt(2) / vmcnt(0), which in fact wait for the h' stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    LAT2_STAMP(step, 2);
    // the acquire's L1 invalidate, hoisted in front of the epilogue (see the first form): no load of h is issued between here and the group barrier
    if (threadIdx.x == 0) asm volatile("buffer_inv sc1" ::: "memory");
    // ---- epilogue: h' = tanh(c1[i] a + bias_t[i - 1]) in bf16, 2 passes of 32 rows through the wave's LDS slab (interior tiles: no bounds branch --
    // a branch per store splits the loop into basic blocks at whose heads the compiler re-waits vmcnt(0), i.e. for the previous store)
    {
      const float alpha = coef[2 * i];
      const float4 bb = *reinterpret_cast<const float4*>(bias_t + (long long)(i - 1) * ldbt + n0 + wc * 128 + (lane & 31) * 4);
      const __bf16* const sdst = hbuf[src ^ 1] + (long long)(m0 + wr * 64) * ldh + n0 + wc * 128;  // (wave-uniform)
      const unsigned voff = ((unsigned)(lane >> 5) * (unsigned)ldh + (unsigned)(lane & 31) * 4u) * 2u;  // the lane's bytes from there: row lane / 32, 4 columns
      if (interior) v2_epilogue<WT, false>(acc, slab, alpha, bb, sdst, voff, (unsigned)ldh, 0, q, fr, lane);
      else v2_epilogue<WT, true>(acc, slab, alpha, bb, sdst, voff, (unsigned)ldh, rows - (m0 + wr * 64), q, fr, lane);
    }
    LAT2_STAMP(step, 3);
    if (step + 1 == T - 1) break;  // h_0 is written; nothing follows in this kernel
    const float sc = coef[2 * i + 1];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const float gn = gp[n * 16];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][n][r] = fmaf(sc, acc[mt][n][r], gn);
    }
    // ---- group barrier (as in the first form)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LAT2_STAMP(step, 4);
    if (threadIdx.x == 0) {
      if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (step + 1) * nt;
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > LATENT_SPIN_MAX) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      if (step == 0) {
        const int mask = __hip_atomic_load(Lb.flags + 256 + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mask & (mask - 1)) __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      LAT2_STAMP(step, 5);
    }
    __syncthreads();
  }
}

static void latent_chain_attrs() {
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_8ph_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_8ph_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_8ph_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_8ph_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_v2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(latent_chain_v2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
    return true;
  }();
  (void)once;
}

// CUs the persistent kernel may count on, per device: 0 when a workgroup of it does not fit a CU at all (occupancy query), else the device's CU count
// (DMM_LATENT_CUS overrides it: a CU-masked or partitioned device the runtime does not describe).
static int latent_chain_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  static std::atomic<int> cached[64];
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v) return v - 1;
  latent_chain_attrs();
  int n = 0, occ = 0;
  if (const char* e = getenv("DMM_LATENT_CUS")) n = atoi(e);
  else if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 0) n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(latent_chain_8ph_kernel<true, true>), 512, P8_LDS) != hipSuccess || occ < 1) n = 0;
  cached[dev].store(n + 1, std::memory_order_relaxed);
  return n;
}

// The group barrier needs all H / 256 member workgroups of the 8 groups of a launch slice co-resident: 8 * H / 256 CUs at least.
bool latent_chain_supported(int H) { return H >= 256 && H % 256 == 0 && H <= 2048 && latent_chain_cus() >= 8 * (H / 256); }

// Workgroups of the second form a CU holds at once (occupancy query; 0 or 1: the first form runs)
static int latent_chain_v2_per_cu() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  static std::atomic<int> cached[64];
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v) return v - 1;
  latent_chain_attrs();
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(latent_chain_v2_kernel<true>), 256, V2_LDS) != hipSuccess || occ < 0) occ = 0;
  cached[dev].store(occ + 1, std::memory_order_relaxed);
  return occ;
}

// part < 0: every launch of the chain; part >= 0: only that launch (groups part * per .. ): the host then places other work between two launches
// (DiffusionEngine.rebuild_edges runs the finished modalities' last GEMM and top-k under the NEXT launch, which leaves CUs idle).
// latent_chain_groups_per_launch: `per`, 0 if the persistent form cannot run.
static int latent_chain_per(int H) {
  static const int form = [] { const char* e = getenv("DMM_LATENT_FORM"); return e ? atoi(e) : 1; }();
  const int cus = latent_chain_cus(), nt = H / 256;
  if (nt < 1 || cus < 8 * nt) return 0;
  if (form == 2 && latent_chain_v2_per_cu() >= 2) return 0;  // (the second form keeps its own slicing: no per-launch calls)
  return (cus / nt) & ~7;
}
int latent_chain_groups_per_launch(int H) { return latent_chain_supported(H) ? latent_chain_per(H) : 0; }

int launch_latent_chain_8ph(const LatentBatch& base, int n_mod, hipStream_t s, int part) {
  latent_chain_attrs();
  // write-through h' stores and no release fence (default; rebuild 5.57 -> 5.20 ms at the bench shape) or plain stores + agent release (0)
  static const int wt = [] { const char* e = getenv("DMM_LATENT_WT"); return e ? atoi(e) : 1; }();
  static const int early = [] { const char* e = getenv("DMM_LATENT_EARLY_INV"); return e ? atoi(e) : 1; }();
  // DMM_LATENT_FORM=2 selects the second form (two 128-row workgroups per CU).  It is correct (same tests) but, as measured in round 4
  // (profiles/r4_chain_forms.txt), slower: a step takes 41-47 us against 31 -- alone on a CU its single wave per SIMD runs the main loop at 20 us
  // for HALF a tile (reads, waits and MFMAs of a stage serialise: the fragment registers cannot be double-buffered next to 128 accumulators within
  // 256 registers) and its epilogue at 14 us; sharing the CU recovers part of it but not enough.  Kept for the record; the first form is the default.
  static const int form = [] { const char* e = getenv("DMM_LATENT_FORM"); return e ? atoi(e) : 1; }();
  static const int contig = [] { const char* e = getenv("DMM_LATENT_CONTIG"); return e ? atoi(e) : 1; }();
  const int cus = latent_chain_cus();
  const int nt = base.H / 256;
  if (cus < 8 * nt) return -1;  // the groups of a launch could not all be resident: the caller runs one launch per step instead
  if (form == 2 && latent_chain_v2_per_cu() >= 2) {
    static const int per_env = [] { const char* e = getenv("DMM_LATENT_V2_GROUPS"); return e ? atoi(e) : 0; }();  // (experiments: fewer groups per launch)
    const int per = per_env > 0 ? std::min(per_env & ~7, (2 * cus / nt) & ~7) : std::min(256, (2 * cus / nt) & ~7);  // groups per launch: every member workgroup resident, two to a CU
    static const int phase = [] { const char* e = getenv("DMM_LATENT_PHASE"); return e ? atoi(e) : 900; }();  // x 10 ns
    LatentBatch b = base;
    b.rbs = (base.rows + 127) / 128;
    b.first_wave = cus / 8;
    b.phase_ticks = phase;
    b.contig = contig;
    const int total = n_mod * b.rbs;
    for (int g0 = 0; g0 < total; g0 += per) {
      b.g0 = g0;
      b.n_groups = std::min(per, total - g0);
      hipError_t e = hipMemsetAsync(b.flags, 0, sizeof(int) * 512, s);
      if (e != hipSuccess) return (int)e;
      const dim3 grid(8 * nt * ((b.n_groups + 7) / 8));
      if (wt) hipLaunchKernelGGL(latent_chain_v2_kernel<true>, grid, dim3(256), V2_LDS, s, b);
      else hipLaunchKernelGGL(latent_chain_v2_kernel<false>, grid, dim3(256), V2_LDS, s, b);
    }
    return 0;
  }
  const int per = (cus / nt) & ~7;  // groups per launch: every member workgroup gets a CU of its own
  const int total = n_mod * base.rbs;
  if (part >= 0 && (long long)part * per >= total) return -2;
  for (int g0 = part >= 0 ? part * per : 0; g0 < total; g0 += per) {
    if (part >= 0 && g0 != part * per) break;
    LatentBatch b = base;
    b.g0 = g0;
    b.n_groups = std::min(per, total - g0);
    b.contig = contig;
    hipError_t e = hipMemsetAsync(b.flags, 0, sizeof(int) * 512, s);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(8 * nt * ((b.n_groups + 7) / 8));
    if (wt && early) hipLaunchKernelGGL((latent_chain_8ph_kernel<true, true>), grid, dim3(512), P8_LDS, s, b);
    else if (wt) hipLaunchKernelGGL((latent_chain_8ph_kernel<true, false>), grid, dim3(512), P8_LDS, s, b);
    else if (early) hipLaunchKernelGGL((latent_chain_8ph_kernel<false, true>), grid, dim3(512), P8_LDS, s, b);
    else hipLaunchKernelGGL((latent_chain_8ph_kernel<false, false>), grid, dim3(512), P8_LDS, s, b);
  }
  return 0;
}

template <int EPI>
static void p8_launch(const GemmBf16Batch& b, hipStream_t s) {
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt_8ph_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL(gemm_bf16_nt_8ph_kernel<EPI>, dim3(b.p[0].mt * b.p[0].nt, b.n), dim3(512), P8_LDS, s, b);
}

void launch_gemm_bf16_8ph(const GemmBf16Batch& b, hipStream_t s) {
  int epi = p8_pick_epi(b.p[0]);
  for (int i = 1; i < b.n; ++i)
    if (p8_pick_epi(b.p[i]) != epi) epi = EPI_GENERIC;  // mixed batches take the run-time dispatch
  switch (epi) {
    case EPI_TANH_CB: p8_launch<EPI_TANH_CB>(b, s); break;
    case EPI_R32_C_CB: p8_launch<EPI_R32_C_CB>(b, s); break;
    case EPI_R16_CB: p8_launch<EPI_R16_CB>(b, s); break;
    case EPI_R16_C: p8_launch<EPI_R16_C>(b, s); break;
    case EPI_TANH_C_CB: p8_launch<EPI_TANH_C_CB>(b, s); break;
    case EPI_C: p8_launch<EPI_C>(b, s); break;
    case EPI_CB: p8_launch<EPI_CB>(b, s); break;
    case EPI_C_CB: p8_launch<EPI_C_CB>(b, s); break;
    case EPI_LATENT: p8_launch<EPI_LATENT>(b, s); break;
    case EPI_LATENT_START: p8_launch<EPI_LATENT_START>(b, s); break;
    default: p8_launch<EPI_GENERIC>(b, s); break;
  }
}

}  // namespace dmm

static int latent_chain_call(const dmm_latent_chain_args* a, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                             int flags_len, int part, void* stream);
extern "C" int dmm_latent_chain(const dmm_latent_chain_args* a, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                                int flags_len, void* stream) {
  return latent_chain_call(a, n_mod, rows, H, T, coef, s_init, flags, flags_len, -1, stream);
}
extern "C" int dmm_latent_chain_part(const dmm_latent_chain_args* a, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                                     int flags_len, int part, void* stream) {
  DMM_CHECK_ARG(part >= 0, "dmm_latent_chain_part: part must be >= 0");
  return latent_chain_call(a, n_mod, rows, H, T, coef, s_init, flags, flags_len, part, stream);
}
extern "C" int dmm_latent_chain_groups_per_launch(int H) { return dmm::latent_chain_groups_per_launch(H); }
static int latent_chain_call(const dmm_latent_chain_args* a, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                             int flags_len, int part, void* stream) {
  using namespace dmm;
  DMM_CHECK_ARG(a && coef && flags && n_mod >= 1 && n_mod <= GB_MAX_BATCH && rows > 0 && T >= 1, "dmm_latent_chain: bad arguments");
  DMM_CHECK_ARG(H >= 256 && H % 256 == 0 && H <= 2048, "dmm_latent_chain: H=%d must be a multiple of 256 in 256..2048", H);
  DMM_CHECK_ARG(flags_len >= 514, "dmm_latent_chain: flags needs 514 int32 entries (256 group counters, 256 placement masks, a spare, the error word)");
  if (T == 1) return 0;  // h_0 = h_{T-1}: nothing to do
  LatentBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < n_mod; ++i) {
    DMM_CHECK_ARG(a[i].G && a[i].g && a[i].z && a[i].h_a && a[i].h_b && a[i].bias_t, "dmm_latent_chain: null operand (problem %d)", i);
    DMM_CHECK_ARG(a[i].ldg >= H && a[i].ldg % 8 == 0 && a[i].ldh >= H && a[i].ldh % 8 == 0 && a[i].ldz >= H && a[i].ldbt >= H && a[i].ldbt % 4 == 0 &&
                      (((uintptr_t)a[i].G | (uintptr_t)a[i].h_a | (uintptr_t)a[i].h_b | (uintptr_t)a[i].bias_t) & 15) == 0,
                  "dmm_latent_chain: G / h / bias_t must be 16-byte aligned with leading dimensions >= H (bf16: multiples of 8, bias_t: of 4)");
    LatentP& P = b.p[i];
    P.G = (const __bf16*)a[i].G; P.ldg = a[i].ldg; P.g = a[i].g; P.z = a[i].z; P.ldz = a[i].ldz;
    P.ha = (__bf16*)a[i].h_a; P.hb = (__bf16*)a[i].h_b; P.ldh = a[i].ldh; P.bias_t = a[i].bias_t; P.ldbt = a[i].ldbt;
  }
  b.coef = coef; b.flags = flags; b.s_init = s_init; b.rows = rows; b.H = H; b.T = T; b.rbs = (rows + 255) / 256; b.err_at = flags_len - 1;
  const int rc = launch_latent_chain_8ph(b, n_mod, (hipStream_t)stream, part);
  if (rc == -2) { dmm::set_error("dmm_latent_chain_part: part %d is past the chain's last launch", part); return DMM_ERR_INVALID_ARG; }
  if (rc == -1) {
    dmm::set_error("dmm_latent_chain: this device cannot keep the %d workgroups of a launch slice co-resident (dmm_latent_chain_supported(%d) == 0): "
                   "run the chain as one dmm_gemm_bf16_nt launch per step", 8 * (H / 256), H);
    return DMM_ERR_UNSUPPORTED;
  }
  if (rc) { dmm::set_error("dmm_latent_chain: hipMemsetAsync failed (%d)", rc); return rc; }
  DMM_LAUNCH_CHECK("dmm_latent_chain");
  return 0;
}

extern "C" int dmm_latent_chain_supported(int H) { return dmm::latent_chain_supported(H) ? 1 : 0; }

#ifdef P8_TIMING
extern "C" int dmm_debug_p8_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dmm::p8_stamps), sizeof(unsigned long long) * n);
}
extern "C" int dmm_debug_latent_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dmm::latent_stamps), sizeof(unsigned long long) * n);
}
#endif
