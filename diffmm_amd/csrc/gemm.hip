// Dense GEMM on the CDNA4 matrix cores: C = post(act(alpha * A B + bias)), any of NN / NT / TN through strides.
//
// Replaces nn.Linear / torch.mm on the hot path: the Denoise MLP and its gate (Model.py:205-215), the modality
// projections (Model.py:84-85,103), the posterior mean (Model.py:375, as an epilogue), the preference
// similarity products (Model.py:416-417) and all of their backward products.
//
// Structure (wave = 64, v_mfma_f32_32x32x16_bf16):
//   * 128 x 128 output tile per 256-thread workgroup; 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles
//     (64 fp32 accumulator registers per lane);
//   * operands live in HBM as fp32 and are rounded to bf16 while being staged through registers into LDS
//     (global_load_dwordx4 -> v_cvt_pk_bf16_f32 -> ds_write_b64); PREC = 3 additionally stages the rounding
//     residual (x - bf16(x)) and issues hi*hi + hi*lo + lo*hi, which recovers ~fp32 accuracy at 1/3 rate;
//   * LDS tiles are [128 rows][32 k] bf16 with an 80-byte row pitch: the 16-lane groups of ds_read_b128 then hit
//     16 distinct 16-byte slots (5 is coprime to 16) -> conflict-free fragment reads;
//   * the next K tile's global loads are issued before the MFMAs of the current one (register prefetch), so HBM/L2
//     latency hides under the matrix work; 4 workgroups per CU (<= 128 VGPRs, 20-40 KiB LDS) hide the barriers.
#include <algorithm>

#include "common.h"

namespace dmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK * 2 + 16;  // bytes per LDS tile row

struct GemmP {
  const float* A; long long sam, sak;
  const float* B; long long sbk, sbn;
  float* C; long long ldc;
  const float* bias;
  const float* R; long long ldr;
  int M, N, K;
  float alpha, e1, e2;
  int act, split_k, accumulate;
  int vecA, vecB;  // 16-byte vector loads legal for this operand
  int k_tiles_per_slice;
};

// One operand tile (128 rows x 32 k) held in registers between the global load and the LDS write.
struct Stage {
  float4 v[4];
};

// KC = true : the operand is k-contiguous (element (r,k) at base + r*ld + k)   -> float4 along k
// KC = false: the operand is row-contiguous (element (r,k) at base + k*ld + r) -> float4 along rows
template <bool KC>
__device__ __forceinline__ void load_tile(Stage& st, const float* __restrict__ base, long long ld, int row0, int rows, int k0, int kend,
                                          bool vec) {
  const int t = threadIdx.x;
  if (KC) {
    const int q = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + r0 + 32 * i, k = k0 + 4 * q;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && k < kend) {
        const float* p = base + (long long)r * ld + k;
        if (vec && k + 3 < kend) {
          x = *reinterpret_cast<const float4*>(p);
        } else {
          x.x = p[0];
          if (k + 1 < kend) x.y = p[1];
          if (k + 2 < kend) x.z = p[2];
          if (k + 3 < kend) x.w = p[3];
        }
      }
      st.v[i] = x;
    }
  } else {
    const int q = t & 31, kk0 = t >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kk0 + 8 * i, r = row0 + 4 * q;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend && r < rows) {
        const float* p = base + (long long)k * ld + r;
        if (vec && r + 3 < rows) {
          x = *reinterpret_cast<const float4*>(p);
        } else {
          x.x = p[0];
          if (r + 1 < rows) x.y = p[1];
          if (r + 2 < rows) x.z = p[2];
          if (r + 3 < rows) x.w = p[3];
        }
      }
      st.v[i] = x;
    }
  }
}

template <bool KC, int PREC>
__device__ __forceinline__ void store_tile(const Stage& st, char* hi, char* lo) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float f[4] = {st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w};
    __bf16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (__bf16)f[j];
      if (PREC == 3) l[j] = (__bf16)(f[j] - (float)h[j]);
    }
    if (KC) {
      const int q = t & 7, r = (t >> 3) + 32 * i;
      bf16x4 hv = {h[0], h[1], h[2], h[3]};
      *reinterpret_cast<bf16x4*>(hi + r * PITCH + q * 8) = hv;
      if (PREC == 3) {
        bf16x4 lv = {l[0], l[1], l[2], l[3]};
        *reinterpret_cast<bf16x4*>(lo + r * PITCH + q * 8) = lv;
      }
    } else {
      const int q = t & 31, k = (t >> 5) + 8 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<__bf16*>(hi + (4 * q + j) * PITCH + k * 2) = h[j];
        if (PREC == 3) *reinterpret_cast<__bf16*>(lo + (4 * q + j) * PITCH + k * 2) = l[j];
      }
    }
  }
}

// Up to GEMM_MAX_BATCH problems of one operand layout in one launch (the modalities' small projections: each alone fills a fifth
// of the chip).  blockIdx.y runs over the problems' row tiles back to back; grid.x / grid.z are the largest column-tile / K-slice
// counts, workgroups beyond a problem's own counts leave at once.
constexpr int GEMM_MAX_BATCH = 4;
struct GemmBatchP {
  GemmP p[GEMM_MAX_BATCH];
  int ybegin[GEMM_MAX_BATCH + 1];
  int n;
};

template <bool AKC, bool BKC, int PREC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmBatchP Bt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = BM * PITCH;  // 10240 bytes
  char* a_hi = smem;
  char* b_hi = smem + TILE;
  char* a_lo = smem + 2 * TILE;
  char* b_lo = smem + 3 * TILE;

  int pi = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_BATCH; ++i)
    if (i < Bt.n && (int)blockIdx.y >= Bt.ybegin[i]) pi = i;
  const GemmP& P = Bt.p[pi];
  const int m0 = ((int)blockIdx.y - Bt.ybegin[pi]) * BM, n0 = blockIdx.x * BN;
  const int total_tiles = (P.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * P.k_tiles_per_slice;
  const int kt_end = min(total_tiles, kt_begin + P.k_tiles_per_slice);
  if (n0 >= P.N || (blockIdx.z > 0 && kt_begin >= kt_end)) return;  // (uniform per workgroup: before any barrier)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A(m,k): k-contiguous when sak == 1 (ld = sam), else row-contiguous (ld = sak).  B(k,n) likewise with n as "row".
  const long long lda = AKC ? P.sam : P.sak, ldb = BKC ? P.sbn : P.sbk;
  Stage sa, sb;
  if (kt_begin < kt_end) {
    load_tile<AKC>(sa, P.A, lda, m0, P.M, kt_begin * BK, P.K, P.vecA);
    load_tile<BKC>(sb, P.B, ldb, n0, P.N, kt_begin * BK, P.K, P.vecB);
  }
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    store_tile<AKC, PREC>(sa, a_hi, a_lo);
    store_tile<BKC, PREC>(sb, b_hi, b_lo);
    __syncthreads();
    if (kt + 1 < kt_end) {  // prefetch: in flight while the MFMAs below run
      load_tile<AKC>(sa, P.A, lda, m0, P.M, (kt + 1) * BK, P.K, P.vecA);
      load_tile<BKC>(sb, P.B, ldb, n0, P.N, (kt + 1) * BK, P.K, P.vecB);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int koff = (ks * 16 + fh * 8) * 2;
      bf16x8 fa[2], fb[2], la[2], lb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(a_hi + (wm * 64 + i * 32 + fr) * PITCH + koff);
        fb[i] = *reinterpret_cast<const bf16x8*>(b_hi + (wn * 64 + i * 32 + fr) * PITCH + koff);
        if (PREC == 3) {
          la[i] = *reinterpret_cast<const bf16x8*>(a_lo + (wm * 64 + i * 32 + fr) * PITCH + koff);
          lb[i] = *reinterpret_cast<const bf16x8*>(b_lo + (wn * 64 + i * 32 + fr) * PITCH + koff);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (PREC == 3) {  // small terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[i], fb[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], lb[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds column n = fr of each 32x32 tile, rows (reg&3) + 8*(reg>>2) + 4*fh ----
  // Every operand is copied out of the kernel-argument struct once (read through the reference, each use was a scalar load + wait), the bias
  // and -- per tile -- the residual / accumulate-into values are loaded before the tile's stores on clamped addresses, and the store kind is
  // chosen outside the element loop: a load inside the per-element bounds branch made the compiler wait vmcnt(0) -- i.e. for the previous
  // element's store -- 192 times per lane (the epilogue was a chain of 192 store round trips).
  const bool first_slice = blockIdx.z == 0;
  float* const C = P.C;
  const float* const Rp = P.R;
  const float* const bias = P.bias;
  const long long ldc = P.ldc, ldr = P.ldr;
  const int M = P.M, N = P.N, act = P.act;
  const bool atomic = P.split_k > 1, accum = P.accumulate != 0;
  const float alpha = P.alpha, e1 = P.e1, e2 = P.e2;
  float bn[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + fr;
    bn[j] = (bias && first_slice) ? bias[n < N ? n : N - 1] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + fr;
      if (n >= N) continue;
      const int mb = m0 + wm * 64 + i * 32 + 4 * fh;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = alpha * acc[i][j][r] + bn[j];
      if (act == DMM_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = tanhf(v[r]);
      } else if (act == DMM_ACT_SIGMOID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
      }
      if (Rp) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          rv[r] = Rp[(long long)(m < M ? m : M - 1) * ldr + n];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = e1 * v[r] + e2 * rv[r];
      }
      if (atomic) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < M) atomicAdd(C + (long long)m * ldc + n, v[r]);
        }
      } else if (accum) {
        float cv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          cv[r] = C[(long long)(m < M ? m : M - 1) * ldc + n];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < M) C[(long long)m * ldc + n] = cv[r] + v[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < M) C[(long long)m * ldc + n] = v[r];
        }
      }
    }
}

template <bool AKC, bool BKC>
static void launch_prec(const GemmBatchP& P, dim3 grid, hipStream_t s, int prec) {
  if (prec == 3)
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, 3>), grid, dim3(256), 4 * BM * PITCH, s, P);
  else
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, 1>), grid, dim3(256), 2 * BM * PITCH, s, P);
}

}  // namespace dmm
using namespace dmm;

static int gemm_fill(const dmm_gemm_args* a, GemmP& P, bool& akc, bool& bkc, int& split) {
  DMM_CHECK_ARG(a && a->A && a->B && a->C, "dmm_gemm: null operand");
  DMM_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "dmm_gemm: bad shape %d x %d x %d", a->M, a->N, a->K);
  DMM_CHECK_ARG(a->precision == 1 || a->precision == 3, "dmm_gemm: precision must be 1 or 3");
  DMM_CHECK_ARG(a->sak == 1 || a->sam == 1, "dmm_gemm: A must be contiguous along m or k");
  DMM_CHECK_ARG(a->sbk == 1 || a->sbn == 1, "dmm_gemm: B must be contiguous along k or n");
  DMM_CHECK_ARG(a->split_k >= 1, "dmm_gemm: split_k >= 1");
  DMM_CHECK_ARG(a->split_k == 1 || (a->act == DMM_ACT_NONE && !a->R && !a->accumulate), "dmm_gemm: split_k > 1 needs act=NONE, R=NULL, accumulate=0");
  P.A = a->A; P.sam = a->sam; P.sak = a->sak;
  P.B = a->B; P.sbk = a->sbk; P.sbn = a->sbn;
  P.C = a->C; P.ldc = a->ldc; P.bias = a->bias; P.R = a->R; P.ldr = a->ldr;
  P.M = a->M; P.N = a->N; P.K = a->K;
  P.alpha = a->alpha; P.e1 = a->e1; P.e2 = a->e2;
  P.act = a->act; P.split_k = a->split_k; P.accumulate = a->accumulate;
  akc = a->sak == 1 && !(a->sam == 1 && a->M > 1 && a->K == 1);
  bkc = a->sbk == 1 && !(a->sbn == 1 && a->N > 1 && a->K == 1);
  const long long lda = akc ? a->sam : a->sak, ldb = bkc ? a->sbn : a->sbk;
  P.vecA = (((uintptr_t)a->A & 15) == 0) && (lda % 4 == 0);
  P.vecB = (((uintptr_t)a->B & 15) == 0) && (ldb % 4 == 0);
  const int tiles = (a->K + BK - 1) / BK;
  split = a->split_k > tiles ? tiles : a->split_k;
  P.k_tiles_per_slice = (tiles + split - 1) / split;
  split = (tiles + P.k_tiles_per_slice - 1) / P.k_tiles_per_slice;
  P.split_k = a->split_k > 1 ? (split > 1 ? split : 2) : 1;  // keep the atomic epilogue whenever the caller pre-zeroed C
  return 0;
}

extern "C" int dmm_gemm_batched(const dmm_gemm_args* a, int n, void* stream) {
  DMM_CHECK_ARG(a && n >= 1 && n <= GEMM_MAX_BATCH, "dmm_gemm_batched: 1..%d problems", GEMM_MAX_BATCH);
  GemmBatchP Bt;
  memset(&Bt, 0, sizeof(Bt));
  Bt.n = n;
  bool akc0 = false, bkc0 = false;
  int y = 0, gx = 0, gz = 1;
  for (int i = 0; i < n; ++i) {
    bool akc, bkc;
    int split;
    if (int rc = gemm_fill(a + i, Bt.p[i], akc, bkc, split)) return rc;
    if (i == 0) { akc0 = akc; bkc0 = bkc; }
    DMM_CHECK_ARG(akc == akc0 && bkc == bkc0 && a[i].precision == a[0].precision, "dmm_gemm_batched: problem %d differs in operand layout or precision", i);
    Bt.ybegin[i] = y;
    y += (int)cdiv(a[i].M, BM);
    gx = std::max(gx, (int)cdiv(a[i].N, BN));
    gz = std::max(gz, split);
  }
  for (int i = n; i <= GEMM_MAX_BATCH; ++i) Bt.ybegin[i] = y;
  dim3 grid(gx, y, gz);
  hipStream_t s = (hipStream_t)stream;
  if (akc0 && bkc0) launch_prec<true, true>(Bt, grid, s, a[0].precision);
  else if (akc0 && !bkc0) launch_prec<true, false>(Bt, grid, s, a[0].precision);
  else if (!akc0 && bkc0) launch_prec<false, true>(Bt, grid, s, a[0].precision);
  else launch_prec<false, false>(Bt, grid, s, a[0].precision);
  DMM_LAUNCH_CHECK("dmm_gemm");
  return 0;
}

extern "C" int dmm_gemm(const dmm_gemm_args* a, void* stream) { return dmm_gemm_batched(a, 1, stream); }
