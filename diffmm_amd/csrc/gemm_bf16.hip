// bf16 x bf16 -> fp32 NT GEMM for the reverse diffusion chain and the diffusion training step: C = post(act(alpha * A B^T + bias)).
// This file: the host entry points, the cast / transpose helpers and the REGISTER-STAGED tiles used for small launches and for
// K % 64 != 0; launches with K % 64 == 0 and >= 64 tiles of 256 x 256 go to the LDS-DMA tile in gemm_bf16_8ph.hip.
//   A [M, K] bf16 row-major (activations), B [N, K] bf16 row-major (nn.Linear weight layout), K % 8 == 0.
// The chain (GaussianDiffusion.generate_view -> p_mean_variance -> Denoise.forward, Model.py:300-322,357-378,210-218) issues
// two of these per timestep, user batch and modality: T * ceil(U/B) * M pairs per epoch -- the FLOP-dominant kernel of the
// metric's T = 40 configuration.  Weights are cast to bf16 once per epoch, activations are produced in bf16 by the previous
// GEMM's epilogue, so staging is a pure 16-byte copy:
//   global_load_dwordx4 -> registers (issued one K tile ahead, in flight under the MFMAs) -> ds_write_b128 into the other
//   LDS buffer -> one barrier per K tile (BK = 64: 16 / 8 MFMA 32x32x16 per wave between barriers).
// LDS tiles are [rows][64 bf16] = 128-byte rows; 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7), which puts the
// 16 rows of every ds_read_b128 lane group on 16 distinct slots of the 256-byte bank row (conflict-free fragment reads).
// Epilogue per element: v = act(alpha*acc + bias[n]); v = e1*v + e2*R[m,n]; written as fp32 (C) and/or bf16 (Cb) -- the
// posterior mean c1*pred + c2*x_t (Model.py:375) and the next step's bf16 operand come out of the same pass.
#include <stdlib.h>

#include "gemm_bf16.h"

namespace dmm {

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ROWS x 64 bf16 tile, NTHR threads: thread t moves the 16-byte chunk (t & 7) of rows (t >> 3) + (NTHR / 8) * i
// `edge` is workgroup-uniform: true when the tile can reach past M / N / K.  Interior workgroups (almost all) move raw registers
// from the load to the LDS write -- nothing consumes the loaded values early, so they stay in flight until that write.  Edge
// workgroups read clamped (valid) addresses and zero the out-of-range chunks with an AND mask (no branch around any load).
template <int ROWS, int NTHR>
__device__ __forceinline__ void g_load(uint4 (&v)[ROWS * 8 / NTHR], const __bf16* __restrict__ base, long long ld, int row0, int rows, int k0, int K,
                                       bool edge) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int k = k0 + c * 8;
  if (!edge) {
    const int kc = k < K ? k : 0;  // only prefetches past the last tile take the clamp; their data is never used
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NTHR; ++i) v[i] = *reinterpret_cast<const uint4*>(base + (long long)(row0 + r0 + (NTHR / 8) * i) * ld + kc);
    return;
  }
  const bool kok = k < K;
  const int kc = kok ? k : 0;
#pragma unroll
  for (int i = 0; i < ROWS * 8 / NTHR; ++i) {
    const int r = row0 + r0 + (NTHR / 8) * i;
    const uint4 x = *reinterpret_cast<const uint4*>(base + (long long)(r < rows ? r : rows - 1) * ld + kc);
    const unsigned msk = (kok && r < rows) ? 0xffffffffu : 0u;
    v[i] = make_uint4(x.x & msk, x.y & msk, x.z & msk, x.w & msk);
  }
}
template <int ROWS, int NTHR>
__device__ __forceinline__ void s_store(const uint4 (&v)[ROWS * 8 / NTHR], char* tile) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < ROWS * 8 / NTHR; ++i) *reinterpret_cast<uint4*>(tile + swz(r0 + (NTHR / 8) * i, c)) = v[i];
}

// One K tile (BK = 64 = 4 MFMA k-steps) of a wave's 64 x (32 NT) output: the fragments of k-step ks+1 are read from LDS while
// the MFMAs of k-step ks run (two fragment register sets), and the interleave is pinned for the scheduler: without it the
// compiler reuses one register set and parks the wave on lgkmcnt(0) in front of every MFMA group.
template <int NT>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][NT], const char* a_t, const char* b_t, int arow, int brow, int fh) {
  bf16x8 fa[2][2], fb[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(a_t + swz(arow + i * 32, fh));
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(b_t + swz(brow + j * 32, fh));
  __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);  // k-step 0's reads
#pragma unroll
  for (int ks = 0; ks < GB_BK / 16; ++ks) {
    const int cb = ks & 1, nb = cb ^ 1;
    if (ks + 1 < GB_BK / 16) {
      const int chunk = (ks + 1) * 2 + fh;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[nb][i] = *reinterpret_cast<const bf16x8*>(a_t + swz(arow + i * 32, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[nb][j] = *reinterpret_cast<const bf16x8*>(b_t + swz(brow + j * 32, chunk));
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);  // the next step's DS reads first ...
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][i], fb[cb][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 2 * NT, 0);        // ... then this step's MFMAs
  }
}

template <int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmBf16P& Pk, f32x16 (&acc)[2][BN / 64], char* smem, int m0, int n0, int wave, int lane) {
  // Every operand is copied out of the kernel-argument struct once: read through the reference, each use re-fetched its field
  // (scalar load + full wait), which made the epilogue slower than a K = 1024 main loop.
  struct { float* C; __bf16* Cb; const float* R; const __bf16* Rb; const float* bias; const float* bias2; long long ldc, ldcb, ldr, ldrb; int M, N, act; float alpha, e1, e2; } P;
  P.C = Pk.C; P.Cb = Pk.Cb; P.R = Pk.R; P.Rb = Pk.Rb; P.bias = Pk.bias; P.bias2 = Pk.bias2; P.ldc = Pk.ldc; P.ldcb = Pk.ldcb; P.ldr = Pk.ldr; P.ldrb = Pk.ldrb;
  P.M = Pk.M; P.N = Pk.N; P.act = Pk.act; P.alpha = Pk.alpha; P.e1 = Pk.e1; P.e2 = Pk.e2;
  constexpr int WN = BN / 2, NT = WN / 32;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  // ---- epilogue: every wave transposes its accumulators through its OWN LDS slab (the staging buffers are free after the last
  // barrier; 32 rows x WN floats per pass, so no workgroup barrier is needed) and then each lane handles 4 consecutive columns:
  // 16-byte residual loads and fp32 stores, 8-byte bf16 stores.
  float* tile = reinterpret_cast<float*>(smem) + wave * 32 * WN;
  constexpr int C4 = WN / 4;       // float4 chunks per tile row
  constexpr int ITER = 32 * C4 / 64;  // chunks per lane per 32-row pass
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const bool vec_ok = (!P.C || (P.ldc % 4 == 0 && ((uintptr_t)P.C & 15) == 0)) &&
                      (!P.R || (P.ldr % 4 == 0 && ((uintptr_t)P.R & 15) == 0)) && (!P.Cb || (P.ldcb % 4 == 0 && ((uintptr_t)P.Cb & 7) == 0)) &&
                      (!P.Rb || (P.ldrb % 4 == 0 && ((uintptr_t)P.Rb & 7) == 0));
  const bool has_r = P.R || P.Rb;
  // the lane's four columns are the same in every chunk (64 % C4 == 0): their bias values are loaded ONCE, before any store (a bias load
  // per element, inside the bounds branch, made every store wait for the one before it)
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const int nl = n0 + wn * WN + (lane % C4) * 4;
    if (P.bias) {
#pragma unroll
      for (int u = 0; u < 4; ++u) b4[u] = P.bias[nl + u < P.N ? nl + u : P.N - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // residual rows are requested first so their latency overlaps the LDS transpose below
    float4 rr[ITER];
    if (has_r && vec_ok) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = it * 64 + lane;
        const int m = m0 + wm * 64 + i * 32 + idx / C4, n = n0 + wn * WN + (idx % C4) * 4;
        rr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < P.M && n + 3 < P.N) {
          if (P.R) {
            rr[it] = *reinterpret_cast<const float4*>(P.R + (long long)m * P.ldr + n);
          } else {
            const bf16x4 b = *reinterpret_cast<const bf16x4*>(P.Rb + (long long)m * P.ldrb + n);
            rr[it] = make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * fh) * WN + j * 32 + fr] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed (DS ops of a wave complete in order)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / C4, c4 = idx % C4;
      const int m = m0 + wm * 64 + i * 32 + row, n = n0 + wn * WN + c4 * 4;
      if (m >= P.M || n >= P.N) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(tile + row * WN + c4 * 4);
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      const int nv = min(4, P.N - n);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < nv) {
          float x = P.alpha * v[u] + b4[u];
          if (P.act == DMM_ACT_TANH) x = P.C ? tanhf(x) : tanh_fast_bf16(x);  // bf16-only output: two transcendentals instead of ~40 instructions
          else if (P.act == DMM_ACT_SIGMOID) x = 1.0f / (1.0f + expf(-x));
          v[u] = x;
        }
      }
      if (nv == 4 && vec_ok) {
        if (has_r) {
          const float4 r4 = rr[it];
          v[0] = P.e1 * v[0] + P.e2 * r4.x; v[1] = P.e1 * v[1] + P.e2 * r4.y; v[2] = P.e1 * v[2] + P.e2 * r4.z; v[3] = P.e1 * v[3] + P.e2 * r4.w;
        }
        if (P.C) *reinterpret_cast<float4*>(P.C + (long long)m * P.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (P.Cb) {
          if (P.bias2) {  // latent chain: the bf16 output is the next activation tanh(C + bias2)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tanh_fast_bf16(v[u] + P.bias2[n + u]);
          }
          *reinterpret_cast<bf16x4*>(P.Cb + (long long)m * P.ldcb + n) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
      } else {
        for (int u = 0; u < nv; ++u) {
          float x = v[u];
          if (P.R) x = P.e1 * x + P.e2 * P.R[(long long)m * P.ldr + n + u];
          else if (P.Rb) x = P.e1 * x + P.e2 * (float)P.Rb[(long long)m * P.ldrb + n + u];
          if (P.C) P.C[(long long)m * P.ldc + n + u] = x;
          if (P.Cb) P.Cb[(long long)m * P.ldcb + n + u] = (__bf16)(P.bias2 ? tanh_fast_bf16(x + P.bias2[n + u]) : x);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // the pass's LDS reads are done before the next pass overwrites the slab
    __builtin_amdgcn_wave_barrier();
  }
}

// BM x BN output tile; waves are arranged (BM / 64) x 2, every wave owns 64 x (BN / 2) outputs (2 x NT MFMA tiles of 32 x 32).
//   <128, 64>  256 threads, 48 KiB LDS: small problems (most workgroups)
//   <128, 128> 256 threads, 64 KiB LDS
//   <256, 128> 512 threads, 96 KiB LDS, 85 flop per staged byte: keeps a CU's vector-memory path (the limiter of the small
//              tiles: 43 flop/B needs ~95 B/clk/CU) under the MFMA time; one workgroup per CU, two waves per SIMD
template <int BM, int BN>
__global__ __launch_bounds__(BM * 2) void gemm_bf16_nt_kernel(const GemmBf16Batch batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = BM * 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WN = BN / 2;        // columns per wave
  constexpr int NT = WN / 32;       // 32x32 tiles per wave along n (2 or 1)
  const GemmBf16P& P = batch.p[blockIdx.y];
  // Workgroups are dealt round-robin over the 8 XCDs (id % 8 names the XCD group; speed only).  With part != 0 each XCD works
  // through a contiguous range of tiles along the dimension of the LARGER operand, so that operand crosses the fabric once.
  int mtile, ntile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    if (P.part == 1) {
      const int mb = xcd * P.mt / 8, me = (xcd + 1) * P.mt / 8;
      if (slot >= (me - mb) * P.nt) return;
      mtile = mb + slot / P.nt;
      ntile = slot % P.nt;
    } else if (P.part == 2) {
      const int nb = xcd * P.nt / 8, ne = (xcd + 1) * P.nt / 8, cx = ne - nb;
      if (slot >= cx * P.mt) return;
      mtile = slot / cx;
      ntile = nb + slot % cx;
    } else {
      mtile = id / P.nt;
      ntile = id % P.nt;
    }
  }
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
  const int nk = (P.K + GB_BK - 1) / GB_BK;
  const bool edge_a = m0 + BM > P.M || P.K % GB_BK != 0, edge_b = n0 + BN > P.N || P.K % GB_BK != 0;

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline: LDS holds tile kt (two buffers), register set 0 / 1 hold tiles kt+1 / kt+2 while their global loads are
  // in flight -- a load has two tiles of MFMA work (~1000+ cycles) to land before it is written to LDS.  The loop is unrolled by
  // two so that both register sets keep static names (runtime-indexed register arrays would go to scratch).
  uint4 ra0[BM * 8 / NTHR], rb0[BN * 8 / NTHR], ra1[BM * 8 / NTHR], rb1[BN * 8 / NTHR];
  auto compute = [&](const char* a_t, const char* b_t) { mma_tile<NT>(acc, a_t, b_t, wm * 64 + fr, wn * WN + fr, fh); };
  char* const buf0 = smem;
  char* const buf1 = smem + STAGE;
  g_load<BM, NTHR>(ra0, P.A, P.lda, m0, P.M, 0, P.K, edge_a);
  g_load<BN, NTHR>(rb0, P.B, P.ldb, n0, P.N, 0, P.K, edge_b);
  s_store<BM, NTHR>(ra0, buf0);
  s_store<BN, NTHR>(rb0, buf0 + A_BYTES);
  g_load<BM, NTHR>(ra0, P.A, P.lda, m0, P.M, GB_BK, P.K, edge_a);       // tile 1 -> set 0
  g_load<BN, NTHR>(rb0, P.B, P.ldb, n0, P.N, GB_BK, P.K, edge_b);
  g_load<BM, NTHR>(ra1, P.A, P.lda, m0, P.M, 2 * GB_BK, P.K, edge_a);   // tile 2 -> set 1
  g_load<BN, NTHR>(rb1, P.B, P.ldb, n0, P.N, 2 * GB_BK, P.K, edge_b);
  __syncthreads();
  // Straight-line body (no conditional loads: tiles past K read clamped addresses and are zero-filled), so the compiler's
  // s_waitcnt insertion can count the loads of the OTHER register set that stay in flight across the LDS write.
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    // even half: LDS buffer 0 = tile kt; set 0 = tile kt+1, set 1 = tile kt+2 (both may still be in flight)
    compute(buf0, buf0 + A_BYTES);
    s_store<BM, NTHR>(ra0, buf1);               // buffer 1 was last read two barriers ago
    s_store<BN, NTHR>(rb0, buf1 + A_BYTES);
    g_load<BM, NTHR>(ra0, P.A, P.lda, m0, P.M, (kt + 3) * GB_BK, P.K, edge_a);
    g_load<BN, NTHR>(rb0, P.B, P.ldb, n0, P.N, (kt + 3) * GB_BK, P.K, edge_b);
    __syncthreads();
    // odd half: LDS buffer 1 = tile kt+1; set 1 = tile kt+2, set 0 = tile kt+3
    compute(buf1, buf1 + A_BYTES);
    s_store<BM, NTHR>(ra1, buf0);
    s_store<BN, NTHR>(rb1, buf0 + A_BYTES);
    g_load<BM, NTHR>(ra1, P.A, P.lda, m0, P.M, (kt + 4) * GB_BK, P.K, edge_a);
    g_load<BN, NTHR>(rb1, P.B, P.ldb, n0, P.N, (kt + 4) * GB_BK, P.K, edge_b);
    __syncthreads();
  }
  if (kt < nk) compute(buf0, buf0 + A_BYTES);  // odd tile count: the last tile sits in buffer 0
  __syncthreads();  // every wave is done with the staging buffers before the epilogue reuses them

  gemm_epilogue<BN>(P, acc, smem, m0, n0, wave, lane);
}

// fp32 [rows, cols] (ld lds) -> bf16 (ld ldd); columns cols..cols_pad are zero-filled
// lo (optional, same layout as dst): the bf16 of what the first rounding lost, x - float(bf16(x)); hi + lo carries 16 mantissa bits
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, long long lds, __bf16* __restrict__ dst, long long ldd,
                                                         int rows, int cols, int cols_pad, __bf16* __restrict__ lo) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)rows * cols_pad; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
    const float x = c < cols ? src[(long long)r * lds + c] : 0.f;
    const __bf16 hi = (__bf16)x;
    dst[(long long)r * ldd + c] = hi;
    if (lo) lo[(long long)r * ldd + c] = (__bf16)(x - (float)hi);
  }
}

// fp32 [R, C] (ld lds) -> bf16 [C, Rp] (ld ldd): tiled transpose through LDS; destination columns R..Rp are zero-filled so the
// result can be the K-contiguous operand of the NT GEMM with K = Rp (a multiple of 8)
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float* __restrict__ src, long long lds, __bf16* __restrict__ dst,
                                                                   long long ldd, int R, int C, int Rp, __bf16* __restrict__ lo) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? src[(long long)r * lds + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) {
      const float x = tile[tx][ty + 4 * i];
      const __bf16 hi = (__bf16)x;
      dst[(long long)c * ldd + r] = hi;
      if (lo) lo[(long long)c * ldd + r] = (__bf16)(x - (float)hi);
    }
  }
}

// dpre = dh * (1 - h^2) as the layer-1 weight-gradient GEMMs' operands (dmm_tanh_bwd_fused): a 64 x 64 tile per workgroup, blockIdx.z = problem.
struct TanhBwdBatch {
  dmm_tanh_bwd_fused_args p[4];
  int R, C, Rp;
};
__global__ __launch_bounds__(256) void tanh_bwd_fused_kernel(const TanhBwdBatch b) {
  __shared__ float tile[64][65];
  __shared__ float cs[4][64];
  const dmm_tanh_bwd_fused_args& P = b.p[blockIdx.z];
  const int R = b.R, C = b.C, Rp = b.Rp;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  __bf16* const d16 = reinterpret_cast<__bf16*>(P.d16);
  __bf16* const dT = reinterpret_cast<__bf16*>(P.dT16);
  float colpart = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      const float t = P.h[(long long)r * P.ld_h + c];
      v = P.dh[(long long)r * P.ld_dh + c] * (1.0f - t * t);
      d16[(long long)r * P.ld16 + c] = (__bf16)v;
    }
    tile[ty + 4 * i][tx] = v;
    colpart += v;
  }
  cs[ty][tx] = colpart;
  __syncthreads();
  if (ty == 0 && c0 + tx < C && r0 < R) atomicAdd(P.csum + c0 + tx, (cs[0][tx] + cs[1][tx]) + (cs[2][tx] + cs[3][tx]));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) dT[(long long)c * P.ldT + r] = (__bf16)tile[tx][ty + 4 * i];
  }
}

// time embedding of a single timestep (every row of the reverse chain shares it) -> bf16 columns I.. of xb, zero padded to ldx
__global__ __launch_bounds__(256) void temb_bias_table_kernel(const float* __restrict__ w_emb, const float* __restrict__ b_emb, int te,
                                                               const float* __restrict__ w1, long long ldw1, int I, const float* __restrict__ b1,
                                                               int H, float* __restrict__ table) {
  __shared__ float e[32], o[32];
  const int t = blockIdx.x;
  if (threadIdx.x < te) {
    const int k = threadIdx.x, half = te / 2;
    float v = 0.f;
    if (!((te & 1) && k == te - 1)) {
      const float f = expf((-9.210340371976184f * (float)(k < half ? k : k - half)) / (float)half);
      const float ang = (float)t * f;
      v = k < half ? cosf(ang) : sinf(ang);
    }
    e[k] = v;
  }
  __syncthreads();
  if (threadIdx.x < te) {
    float v = b_emb[threadIdx.x];
    for (int k = 0; k < te; ++k) v = fmaf(w_emb[threadIdx.x * te + k], e[k], v);
    o[threadIdx.x] = v;
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += 256) {
    float v = b1[h];
    for (int c = 0; c < te; ++c) v = fmaf(w1[(long long)h * ldw1 + I + c], o[c], v);
    table[(long long)t * H + h] = v;
  }
}

}  // namespace dmm
using namespace dmm;

extern "C" int dmm_temb_bias_table(const float* w_emb, const float* b_emb, int te, const float* w1, int64_t ldw1, int I, const float* b1, int H,
                                   int T, float* table, void* stream) {
  DMM_CHECK_ARG(w_emb && b_emb && w1 && b1 && table && te >= 2 && te <= 32 && H > 0 && T > 0 && ldw1 >= I + te, "dmm_temb_bias_table: bad arguments");
  hipLaunchKernelGGL(temb_bias_table_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, w_emb, b_emb, te, w1, (long long)ldw1, I, b1, H, table);
  DMM_LAUNCH_CHECK("dmm_temb_bias_table");
  return 0;
}

template <int BM, int BN>
static void launch_tile(const GemmBf16Batch& b, int blocks, hipStream_t s) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  static bool once = [] {  // > 64 KiB of dynamic LDS must be opted into
    if (LDS > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN>), dim3(blocks, b.n), dim3(BM * 2), LDS, s, b);
}

extern "C" int dmm_gemm_bf16_nt_batched(const dmm_gemm_bf16_args* args, int n_batch, void* stream) {
  DMM_CHECK_ARG(args && n_batch >= 1 && n_batch <= GB_MAX_BATCH, "dmm_gemm_bf16_nt: 1..%d problems per launch", GB_MAX_BATCH);
  GemmBf16Batch b;
  memset(&b, 0, sizeof(b));
  b.n = n_batch;
  const dmm_gemm_bf16_args* a0 = args;
  for (int i = 0; i < n_batch; ++i) {
    const dmm_gemm_bf16_args* a = args + i;
    DMM_CHECK_ARG(a->A && a->B && (a->C || a->Cb), "dmm_gemm_bf16_nt: null operand (problem %d)", i);
    DMM_CHECK_ARG(!a->bias2 || (a->C && a->Cb), "dmm_gemm_bf16_nt: bias2 needs both outputs (problem %d)", i);
    DMM_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->K % 8 == 0, "dmm_gemm_bf16_nt: K=%d must be a positive multiple of 8", a->K);
    DMM_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && (((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0,
                  "dmm_gemm_bf16_nt: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
    DMM_CHECK_ARG(a->M == a0->M && a->N == a0->N && a->K == a0->K, "dmm_gemm_bf16_nt: batched problems must share one shape");
    GemmBf16P& P = b.p[i];
    P.A = (const __bf16*)a->A; P.lda = a->lda; P.B = (const __bf16*)a->B; P.ldb = a->ldb;
    P.C = a->C; P.ldc = a->ldc; P.Cb = (__bf16*)a->Cb; P.ldcb = a->ldcb;
    P.bias = a->bias; P.bias2 = a->bias2; P.R = a->R; P.ldr = a->ldr; P.Rb = (const __bf16*)a->Rb; P.ldrb = a->ldrb; P.M = a->M; P.N = a->N; P.K = a->K;
    P.alpha = a->alpha; P.e1 = a->e1; P.e2 = a->e2; P.act = a->act;
  }
  static const int force_tile = [] { const char* e = getenv("DMM_GEMM_TILE"); return e ? atoi(e) : 0; }();  // tuning knobs: 64 / 128 / 256
  static const int force_part = [] { const char* e = getenv("DMM_GEMM_PART"); return e ? atoi(e) : -1; }();
  const int M = a0->M, N = a0->N;
  // 256 x 256 LDS-DMA tile (gemm_bf16_8ph.hip): K must be whole 64-wide tiles and the grid should fill at least a quarter of the
  // 256 CUs (the chain runs one launch per modality on concurrent streams; with users sharded over 2 replicas a layer-1 launch has 76
  // tiles -- one round of the LDS-DMA tile is still twice as fast as the register-staged tiles at that size)
  static const int use_8ph = [] { const char* e = getenv("DMM_GEMM_8PH"); return e ? atoi(e) : 1; }();  // 0: register-staged tiles only
  const long long t8 = (long long)cdiv(M, 256) * cdiv(N, 256) * n_batch;
  static const int min_t8 = [] { const char* e = getenv("DMM_GEMM_8PH_MIN_TILES"); return e ? atoi(e) : 64; }();  // (tuning knob)
  if (a0->K % 64 == 0 && (force_tile == 512 || (force_tile == 0 && use_8ph && t8 >= min_t8))) {
    for (int i = 0; i < n_batch; ++i) {
      b.p[i].mt = cdiv(M, 256); b.p[i].nt = cdiv(N, 256); b.p[i].part = force_part >= 0 ? force_part : 2;
    }
    launch_gemm_bf16_8ph(b, (hipStream_t)stream);
    DMM_LAUNCH_CHECK("dmm_gemm_bf16_nt (256x256)");
    return 0;
  }
  const long long t256 = (long long)cdiv(M, 256) * cdiv(N, 128) * n_batch, t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * n_batch;
  int tile = t256 >= 768 ? 256 : (t128 >= 512 ? 128 : 64);  // big tiles only when they still fill the 256 CUs several times over
  // (round 4: taking 128 x 128 for the long-K launches the rule below gives 128 x 64 -- the third diffusion batch's layer-1 / dh products, 216 tiles; the
  // G refresh, 192 -- measured slower or neutral)
  if (force_tile) tile = force_tile;
  const int bm = tile == 256 ? 256 : 128, bn = tile == 64 ? 64 : 128;
  const int mt = cdiv(M, bm), nt = cdiv(N, bn);
  // XCD-contiguous placement pays only when the operands overflow the 256 MiB Infinity Cache (measured: +-3 % on the 70 MB
  // chain shapes, where grid order is kept); then the XCDs split the dimension of the larger operand.
  int part = 0;
  if (((long long)M + N) * a0->K * 2 * n_batch > (192ll << 20)) part = M >= N ? (mt >= 8 ? 1 : 0) : (nt >= 8 ? 2 : 0);
  if (force_part >= 0) part = force_part;
  int blocks = mt * nt;
  if (part == 1) {
    int mx = 0;
    for (int x = 0; x < 8; ++x) mx = max(mx, (x + 1) * mt / 8 - x * mt / 8);
    blocks = 8 * mx * nt;
  } else if (part == 2) {
    int mx = 0;
    for (int x = 0; x < 8; ++x) mx = max(mx, (x + 1) * nt / 8 - x * nt / 8);
    blocks = 8 * mx * mt;
  }
  for (int i = 0; i < n_batch; ++i) {
    b.p[i].mt = mt; b.p[i].nt = nt; b.p[i].part = part;
  }
  hipStream_t s = (hipStream_t)stream;
  if (tile == 256) launch_tile<256, 128>(b, blocks, s);
  else if (tile == 128) launch_tile<128, 128>(b, blocks, s);
  else launch_tile<128, 64>(b, blocks, s);
  DMM_LAUNCH_CHECK("dmm_gemm_bf16_nt");
  return 0;
}

extern "C" int dmm_gemm_bf16_nt(const dmm_gemm_bf16_args* a, void* stream) { return dmm_gemm_bf16_nt_batched(a, 1, stream); }

extern "C" int dmm_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int cols_pad, void* stream) {
  DMM_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols_pad >= cols && ldd >= cols_pad, "dmm_cast_bf16: bad arguments");
  const long long n = (long long)rows * cols_pad;
  const unsigned grid = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (long long)lds, (__bf16*)dst, (long long)ldd, rows, cols, cols_pad,
                     (__bf16*)nullptr);
  DMM_LAUNCH_CHECK("dmm_cast_bf16");
  return 0;
}

extern "C" int dmm_cast_bf16_split(const float* src, int64_t lds, void* hi, void* lo, int64_t ldd, int rows, int cols, int cols_pad, void* stream) {
  DMM_CHECK_ARG(src && hi && lo && rows > 0 && cols > 0 && cols_pad >= cols && ldd >= cols_pad, "dmm_cast_bf16_split: bad arguments");
  const long long n = (long long)rows * cols_pad;
  const unsigned grid = (unsigned)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (long long)lds, (__bf16*)hi, (long long)ldd, rows, cols, cols_pad,
                     (__bf16*)lo);
  DMM_LAUNCH_CHECK("dmm_cast_bf16_split");
  return 0;
}

extern "C" int dmm_transpose_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int rows_pad, void* stream) {
  DMM_CHECK_ARG(src && dst && rows > 0 && cols > 0 && rows_pad >= rows && ldd >= rows_pad, "dmm_transpose_cast_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_cast_bf16_kernel, dim3(cdiv(cols, 64), cdiv(rows_pad, 64)), dim3(256), 0, (hipStream_t)stream, src, (long long)lds,
                     (__bf16*)dst, (long long)ldd, rows, cols, rows_pad, (__bf16*)nullptr);
  DMM_LAUNCH_CHECK("dmm_transpose_cast_bf16");
  return 0;
}

extern "C" int dmm_tanh_bwd_fused(const dmm_tanh_bwd_fused_args* a, int n, int rows, int cols, int rows_pad, void* stream) {
  DMM_CHECK_ARG(a && n >= 1 && n <= 4 && rows > 0 && cols > 0 && rows_pad >= rows && rows_pad % 8 == 0, "dmm_tanh_bwd_fused: bad arguments");
  dmm::TanhBwdBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < n; ++i) {
    DMM_CHECK_ARG(a[i].dh && a[i].h && a[i].d16 && a[i].dT16 && a[i].csum && a[i].ld_dh >= cols && a[i].ld_h >= cols && a[i].ld16 >= cols && a[i].ldT >= rows_pad,
                  "dmm_tanh_bwd_fused: null operand or short leading dimension (problem %d)", i);
    b.p[i] = a[i];
  }
  b.R = rows; b.C = cols; b.Rp = rows_pad;
  hipLaunchKernelGGL(dmm::tanh_bwd_fused_kernel, dim3(dmm::cdiv(cols, 64), dmm::cdiv(rows_pad, 64), n), dim3(256), 0, (hipStream_t)stream, b);
  DMM_LAUNCH_CHECK("dmm_tanh_bwd_fused");
  return 0;
}

extern "C" int dmm_transpose_cast_bf16_split(const float* src, int64_t lds, void* hi, void* lo, int64_t ldd, int rows, int cols, int rows_pad, void* stream) {
  DMM_CHECK_ARG(src && hi && lo && rows > 0 && cols > 0 && rows_pad >= rows && ldd >= rows_pad, "dmm_transpose_cast_bf16_split: bad arguments");
  hipLaunchKernelGGL(transpose_cast_bf16_kernel, dim3(cdiv(cols, 64), cdiv(rows_pad, 64)), dim3(256), 0, (hipStream_t)stream, src, (long long)lds,
                     (__bf16*)hi, (long long)ldd, rows, cols, rows_pad, (__bf16*)lo);
  DMM_LAUNCH_CHECK("dmm_transpose_cast_bf16_split");
  return 0;
}

// out[r] = sum_c W[r, c] x[c]: one workgroup per row (the latent chain's g = W1x b2: an [H, I] matrix against the layer-2 bias)
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, long long ldw, const float* __restrict__ x, float* __restrict__ out, int cols) {
  const float* __restrict__ w = W + (long long)blockIdx.x * ldw;
  float a = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) a += w[c] * x[c];
  __shared__ float part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

extern "C" int dmm_gemv_rows(const float* W, int64_t ldw, const float* x, float* out, int rows, int cols, void* stream) {
  DMM_CHECK_ARG(W && x && out && rows > 0 && cols > 0 && ldw >= cols, "dmm_gemv_rows: bad arguments");
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, W, (long long)ldw, x, out, cols);
  DMM_LAUNCH_CHECK("dmm_gemv_rows");
  return 0;
}

