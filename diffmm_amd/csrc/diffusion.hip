// Diffusion-path kernels that are not GEMMs: q_sample + time embedding (operand builder of Denoise layer 1),
// the diffusion loss with its direct gradients, and the per-user top-k of the rebuild.
// Reference: GaussianDiffusion.forward_cal_xt (Model.py:324-341), Denoise.forward time embedding (Model.py:196-202),
// training_losses (Model.py:407-425), Coach.trainEpoch rebuild loop (Main.py:224-230).
#include "common.h"

namespace dmm {

constexpr int MAX_TE = 32;

__device__ __forceinline__ void time_features(int t, int te, float* e /*[te]*/) {
  const int half = te / 2;
  for (int k = 0; k < half; ++k) {
    const float f = expf((-9.210340371976184f * (float)k) / (float)half);
    const float ang = (float)t * f;
    e[k] = cosf(ang);
    e[half + k] = sinf(ang);
  }
  if (te & 1) e[te - 1] = 0.f;
}

// one workgroup per batch row
__global__ __launch_bounds__(256) void qsample_temb_kernel(const float* __restrict__ x0, long long ldx0, const float* __restrict__ noise,
                                                            const int32_t* __restrict__ t, const float* __restrict__ sqrt_ac,
                                                            const float* __restrict__ sqrt_1mac, const float* __restrict__ w_emb,
                                                            const float* __restrict__ b_emb, int te, float* __restrict__ xcat, long long ldx,
                                                            int I, int draw_noise, unsigned long long seed, unsigned long long offset,
                                                            float* __restrict__ eps_out, const uint32_t* __restrict__ bits, long long ldb) {
  const int b = blockIdx.x;
  const int tb = t[b];
  float* __restrict__ row = xcat + (long long)b * ldx;
  if (x0 || bits) {  // bits: x_0 as one bit per (row, item) -- 0/1 interaction rows read at 1/32 of the bytes (dmm_csr_row_bits)
    const bool has_eps = noise != nullptr || draw_noise;
    const float a = has_eps ? sqrt_ac[tb] : 1.0f, c = has_eps ? sqrt_1mac[tb] : 0.0f;
    const float* __restrict__ xr = x0 ? x0 + (long long)b * ldx0 : nullptr;
    const uint32_t* __restrict__ br = bits ? bits + (long long)b * ldb : nullptr;
    if (noise || !draw_noise) {
      for (int j = threadIdx.x; j < I; j += 256) {
        const float e = noise ? noise[(long long)b * I + j] : 0.f;
        const float xv = br ? (float)((br[j >> 5] >> (j & 31)) & 1u) : xr[j];
        row[j] = a * xv + c * e;
        if (eps_out) eps_out[(long long)b * I + j] = e;
      }
    } else {
      // 4 normals per Philox call: thread handles columns 4q .. 4q+3
      for (int q = threadIdx.x; q * 4 < I; q += 256) {
        const uint4 r = philox4(seed, offset + (unsigned long long)b * ((I + 3) / 4) + q, 0xd1ffu);
        const float2 n0 = box_muller(r.x, r.y), n1 = box_muller(r.z, r.w);
        const float e[4] = {n0.x, n0.y, n1.x, n1.y};
        const uint32_t word = br ? br[q >> 3] >> ((q & 7) * 4) : 0u;  // columns 4q .. 4q+3 sit in one word
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * q + u;
          if (j < I) {
            const float xv = br ? (float)((word >> u) & 1u) : xr[j];
            row[j] = a * xv + c * e[u];
            if (eps_out) eps_out[(long long)b * I + j] = e[u];
          }
        }
      }
    }
  }
  // time embedding -> columns I .. I+te, zero padding up to ldx
  __shared__ float e[MAX_TE];
  if (threadIdx.x == 0) time_features(tb, te, e);
  __syncthreads();
  for (int c = threadIdx.x; I + c < ldx; c += 256) {
    float v = 0.f;
    if (c < te) {
      v = b_emb[c];
      for (int k = 0; k < te; ++k) v = fmaf(w_emb[c * te + k], e[k], v);
    }
    row[I + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// x_0 as CSR rows (the reference keeps a dense U x I matrix on the device, DataHandler.py:128 -- impossible at I = 1e6):
//   csr_dense_rows      out[b, :ld] = 0/1 row of user rows[b]                              (DiffusionData.__getitem__, DataHandler.py:218-224)
//   csr_gather_sum      h[b, :] = act(bias + sum_{j in items(rows[b])} Wt[j, :])            (Denoise layer 1 on x_t = x_0: a gather of W1^T
//                       rows instead of a dense GEMM operand -- the first step of the reverse chain, Model.py:300-322 with sampling_step 0)
//   csr_add_rows        y[b, j] += c for j in items(rows[b]); y16 = bf16(y) there           (that step's posterior-mean residual c2 * x_0)
// One workgroup per row; ptr is the CSR of ALL users (int64), col int32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_dense_rows_kernel(const long long* __restrict__ ptr, const int32_t* __restrict__ col,
                                                             const long long* __restrict__ rows, float* __restrict__ out, long long ld, int I, int col_lo) {
  const int b = blockIdx.x;
  float* __restrict__ o = out + (long long)b * ld;
  for (long long j = threadIdx.x * 4; j < ld; j += 1024) {
    if (j + 4 <= ld) *reinterpret_cast<float4*>(o + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    else for (long long q = j; q < ld; ++q) o[q] = 0.f;
  }
  __syncthreads();
  const long long u = rows[b], beg = ptr[u], end = ptr[u + 1];
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int c = col[e] - col_lo;  // (col_lo: the first item of the column slice this replica holds; 0 = all items)
    if ((unsigned)c < (unsigned)I) o[c] = 1.0f;
  }
}

// (up to 4 problems per launch, blockIdx.y: the same rows against several operand tables -- the M modalities' chain starts -- so that the heavy
// users' long gather loops of one problem run under the others' instead of three latency-bound launches in a row)
struct CsrGatherP { const __bf16* Wt; long long ldw; const float* bias; __bf16* out16; long long ldo; float* z; long long ldz; };
struct CsrGatherBatch { CsrGatherP p[4]; };
template <int ACT>
__global__ __launch_bounds__(256) void csr_gather_sum_kernel(const long long* __restrict__ ptr, const int32_t* __restrict__ col,
                                                             const long long* __restrict__ rows, const CsrGatherBatch batch, int H) {
  const CsrGatherP& P = batch.p[blockIdx.y];
  const __bf16* __restrict__ Wt = P.Wt; const float* __restrict__ bias = P.bias; __bf16* __restrict__ out16 = P.out16; float* __restrict__ z = P.z;
  const long long ldw = P.ldw, ldo = P.ldo, ldz = P.ldz;
  const int b = blockIdx.x;
  const long long u = rows[b], beg = ptr[u], end = ptr[u + 1];
  for (int h0 = threadIdx.x * 4; h0 < H; h0 += 1024) {  // 4 columns (8 bytes of bf16) per thread per pass
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (!z) { a[0] = bias[h0]; a[1] = bias[h0 + 1]; a[2] = bias[h0 + 2]; a[3] = bias[h0 + 3]; }
    long long e = beg;
    for (; e + 8 <= end; e += 8) {  // 8 independent row gathers in flight (a heavy user has hundreds of items: the loop is latency-bound)
      uint2 w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const uint2*>(Wt + (long long)col[e + j] * ldw + h0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[0] += __uint_as_float(w[j].x << 16); a[1] += __uint_as_float(w[j].x & 0xffff0000u);
        a[2] += __uint_as_float(w[j].y << 16); a[3] += __uint_as_float(w[j].y & 0xffff0000u);
      }
    }
    for (; e < end; ++e) {
      const uint2 w = *reinterpret_cast<const uint2*>(Wt + (long long)col[e] * ldw + h0);
      a[0] += __uint_as_float(w.x << 16); a[1] += __uint_as_float(w.x & 0xffff0000u);
      a[2] += __uint_as_float(w.y << 16); a[3] += __uint_as_float(w.y & 0xffff0000u);
    }
    if (z) {  // latent chain start: the bias-free sum is the fp32 state z_T = x_0 W1x^T; the bias joins for the activation only
      *reinterpret_cast<float4*>(z + (long long)b * ldz + h0) = make_float4(a[0], a[1], a[2], a[3]);
      a[0] += bias[h0]; a[1] += bias[h0 + 1]; a[2] += bias[h0 + 2]; a[3] += bias[h0 + 3];
    }
    __bf16 o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (__bf16)(ACT == 1 ? tanhf(a[q]) : a[q]);
    *reinterpret_cast<uint2*>(out16 + (long long)b * ldo + h0) = *reinterpret_cast<const uint2*>(o);
  }
}

__global__ __launch_bounds__(256) void csr_add_rows_kernel(const long long* __restrict__ ptr, const int32_t* __restrict__ col,
                                                           const long long* __restrict__ rows, float c, float* __restrict__ y, long long ldy,
                                                           __bf16* __restrict__ y16, long long ldy16, int I) {
  const int b = blockIdx.x;
  const long long u = rows[b], beg = ptr[u], end = ptr[u + 1];
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int j = col[e];
    if ((unsigned)j >= (unsigned)I) continue;
    // a user's items are distinct: no two threads touch one element.  y == NULL: the row lives in bf16 only (y16 is read and written)
    const float v = (y ? y[(long long)b * ldy + j] : (float)y16[(long long)b * ldy16 + j]) + c;
    if (y) y[(long long)b * ldy + j] = v;
    if (y16) y16[(long long)b * ldy16 + j] = (__bf16)v;
  }
}

// out[b, :] = a[t_b] * x[b, :] + c[t_b] * y[b, :] with per-row timesteps: GaussianDiffusion.forward_cal_xt with explicit noise
// (Model.py:339-341: x = x_0, y = noise, tables sqrt(alpha_bar), sqrt(1 - alpha_bar)) and the posterior mean of p_mean_variance
// (Model.py:375: x = Denoise output, y = x_t, tables posterior_mean_coef1 / coef2) for callers of the reference-named API.
__global__ __launch_bounds__(256) void rows_axpby_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ y, long long ldy,
                                                         const int32_t* __restrict__ t, const float* __restrict__ a, const float* __restrict__ c,
                                                         float* __restrict__ out, long long ldo, int I) {
  const int b = blockIdx.x;
  const float ab = a[t[b]], cb = c[t[b]];
  const float* __restrict__ xr = x + (long long)b * ldx;
  const float* __restrict__ yr = y + (long long)b * ldy;
  float* __restrict__ o = out + (long long)b * ldo;
  for (int j = threadIdx.x; j < I; j += 256) o[j] = ab * xr[j] + cb * yr[j];
}

// Start of a reverse chain with sampling_step > 0 (GaussianDiffusion.generate_view -> forward_cal_xt without a noise argument,
// Model.py:310-313,336-341): x_T = sqrt_ac[s] * x_0 + sqrt_1mac[s] * sign(x_0) * randn / max(||randn_row||, 1e-12) -- the noise is
// row-normalised over ALL I columns and then kept only where x_0 is non-zero.  One workgroup per row; the row's normals are drawn
// twice from the same Philox counters (norm pass, write pass) instead of being stored.  `randn` (optional) replaces the draw.
__global__ __launch_bounds__(256) void noised_start_kernel(const float* __restrict__ x0, long long ldx0, const float* __restrict__ randn, float a, float c,
                                                           float* __restrict__ xT, long long ldx, int I, unsigned long long seed,
                                                           unsigned long long offset) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* __restrict__ xr = x0 + (long long)b * ldx0;
  float* __restrict__ row = xT + (long long)b * ldx;
  const unsigned long long base = offset + (unsigned long long)b * ((I + 3) / 4);
  float ss = 0.f;
  for (int q = threadIdx.x; q * 4 < I; q += 256) {
    float e[4];
    if (randn) {
      for (int u = 0; u < 4; ++u) e[u] = 4 * q + u < I ? randn[(long long)b * I + 4 * q + u] : 0.f;
    } else {
      const uint4 r = philox4(seed, base + q, 0xd1ffu);
      const float2 n0 = box_muller(r.x, r.y), n1 = box_muller(r.z, r.w);
      e[0] = n0.x; e[1] = n0.y; e[2] = n1.x; e[3] = n1.y;
    }
    for (int u = 0; u < 4; ++u)
      if (4 * q + u < I) ss = fmaf(e[u], e[u], ss);
  }
  ss = block_sum_256(ss, red);
  const float scale = c / fmaxf(sqrtf(ss), 1e-12f);
  for (int q = threadIdx.x; q * 4 < I; q += 256) {
    float e[4];
    if (randn) {
      for (int u = 0; u < 4; ++u) e[u] = 4 * q + u < I ? randn[(long long)b * I + 4 * q + u] : 0.f;
    } else {
      const uint4 r = philox4(seed, base + q, 0xd1ffu);
      const float2 n0 = box_muller(r.x, r.y), n1 = box_muller(r.z, r.w);
      e[0] = n0.x; e[1] = n0.y; e[2] = n1.x; e[3] = n1.y;
    }
    for (int u = 0; u < 4; ++u) {
      const int j = 4 * q + u;
      if (j < I) {
        const float x = xr[j];
        const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        row[j] = a * x + scale * sg * e[u];
      }
    }
  }
}

// d w_emb[c][k] += sum_b dtemb[b][c] * e_b[k],  d b_emb[c] += sum_b dtemb[b][c]   (te <= 32).  A workgroup takes TEMB_WG_ROWS batch rows:
// thread = (row lane r, pair (c, k)) with R = 1024 / te^2 row lanes (10 at te = 10: every thread works, where one (c, k) pair per thread left 9 in 10
// idle), each walking the rows r, r + R, ...; the row lanes' partial sums are combined through LDS and ONE fp32 atomic per pair and workgroup goes
// out.  (Round-2 form: 8 rows per workgroup, 512 workgroups at B = 4096, every pair's partial added by atomics -- 56k atomics on 110 addresses:
// 25 us per modality, all contention.  Now 52 workgroups, 5.7k atomics.)
constexpr int TEMB_WG_ROWS = 80;
__global__ __launch_bounds__(1024) void temb_bwd_kernel(const int32_t* __restrict__ t, const float* __restrict__ dxcat, long long ldx, int I,
                                                         int te, int B, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double sw[1024], sb[1024];
  const int pairs = te * te, R = 1024 / pairs;
  const int r = threadIdx.x / pairs, pk = threadIdx.x % pairs;
  const int c = pk / te, k = pk % te;
  const int b0 = blockIdx.x * TEMB_WG_ROWS, b1 = min(B, b0 + TEMB_WG_ROWS);
  double aw = 0.0, ab = 0.0;
  if (r < R) {
    const int half = te / 2;
    const bool is_pad = (te & 1) && k == te - 1;
    const float f = expf((-9.210340371976184f * (float)(k < half ? k : k - half)) / (float)half);
    for (int b = b0 + r; b < b1; b += R) {
      const float ang = (float)t[b] * f;
      const float ek = is_pad ? 0.f : (k < half ? cosf(ang) : sinf(ang));
      const float g = dxcat[(long long)b * ldx + I + c];
      aw += (double)g * (double)ek;
      ab += (double)g;
    }
  }
  sw[threadIdx.x] = aw;
  sb[threadIdx.x] = ab;
  __syncthreads();
  if (r == 0) {  // the row lanes in order: a fixed summation order inside the workgroup
    for (int j = 1; j < R; ++j) { aw += sw[j * pairs + pk]; ab += sb[j * pairs + pk]; }
    atomicAdd(dw + c * te + k, (float)aw);
    if (k == 0) atomicAdd(db + c, (float)ab);
  }
}

// diffusion loss: one workgroup per batch row
__global__ __launch_bounds__(256) void diff_loss_kernel(const float* __restrict__ out, const float* __restrict__ x0, long long ld,
                                                         const double* __restrict__ w, const float* __restrict__ um,
                                                         const float* __restrict__ ui, int I, int d, float sim_weight, double reg_term,
                                                         double gscale, double* __restrict__ loss, float* __restrict__ dout,
                                                         float* __restrict__ dum, const uint32_t* __restrict__ bits, long long ldb,
                                                         const double* __restrict__ gs_dev) {
  if (gs_dev) gscale *= *gs_dev;  // a factor of the gradient scale that only exists on the device (1 / sum of the minibatch's losses)
  __shared__ double smd[4];
  __shared__ float smf[4];
  const int b = blockIdx.x;
  const float* __restrict__ o = out + (long long)b * ld;
  const float* __restrict__ x = x0 ? x0 + (long long)b * ld : nullptr;
  const uint32_t* __restrict__ br = bits ? bits + (long long)b * ldb : nullptr;  // x_0 as a bit mask (dmm_csr_row_bits)
  const double wb = w[b];
  const float gcoef = (float)(gscale * wb * 2.0 / (double)I);
  double a = 0.0;
  if (loss || dout) {  // (loss == NULL and dout == NULL: only d loss / d um is wanted -- the rows are not read at all)
    for (int j = threadIdx.x; j < I; j += 256) {
      const float df = o[j] - (br ? (float)((br[j >> 5] >> (j & 31)) & 1u) : x[j]);
      a += (double)(df * df);
      if (dout) dout[(long long)b * ld + j] = gcoef * df;
    }
  }
  const double sse = block_sum_256d(a, smd);
  // cosine similarity of the two d-vectors
  float dot = 0.f, nu = 0.f, ni = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float p = um[(long long)b * d + c], q = ui[(long long)b * d + c];
    dot += p * q; nu += p * p; ni += q * q;
  }
  dot = block_sum_256(dot, smf);
  nu = sqrtf(block_sum_256(nu, smf));
  ni = sqrtf(block_sum_256(ni, smf));
  const float cu = fmaxf(nu, 1e-8f), ci = fmaxf(ni, 1e-8f);
  const float cosv = dot / (cu * ci);
  if (threadIdx.x == 0 && loss) loss[b] = wb * (sse / (double)I) + (double)sim_weight * (1.0 - (double)cosv) + reg_term;
  if (dum) {
    const float gs = (float)gscale * sim_weight;
    for (int c = threadIdx.x; c < d; c += 256) {
      const float p = um[(long long)b * d + c], q = ui[(long long)b * d + c];
      // d(1 - cos)/d um = -( q / (cu ci) - cos * p / cu^2 )   (the second term vanishes in the clamped branch)
      const float dcos = q / (cu * ci) - (nu > 1e-8f ? cosv * p / (cu * cu) : 0.f);
      dum[(long long)b * d + c] = -gs * dcos;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// per-row top-k (k = the user's train degree): bitwise threshold search on order-preserving keys,
// then an ordered compaction.  Deterministic; ties at the threshold go to the smaller column id.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int block_sum_int(int v, int* sm4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// Rows beyond the register-resident forms (> 20480 columns) are read THREE times: (1) an 11-bit histogram of the keys' top bits (sign, exponent, two
// mantissa bits) finds the bucket holding rank k; (2) that bucket's keys -- a thin slice of the row's upper tail -- are collected in LDS and the
// remaining 21 bits are resolved there; (3) the ordered compaction.  A bucket with more than TOPK_CAND keys (a row concentrated on one value)
// resolves its remaining bits by passes over the row instead.  (The first version read the row six times: four 8-bit passes, a count of the
// keys above the threshold -- which is k minus the threshold's rank among its ties -- and the compaction: 32 -> 19 ms per 16384 x 100k.)
constexpr int TOPK_BINS = 2048, TOPK_CAND = 4096;
struct TopkStreamShared {
  int hist[TOPK_BINS];
  uint32_t cand[TOPK_CAND];
  int wsum[4], sm4[4], wave_off[5];
  int cscan[2][4];  // the compaction's per-wave (above | ties << 16) totals, double-buffered
  uint32_t sel_prefix;
  int sel_k, sel_cnt, n_cand;
};
// bins[0 .. 256 Q): the bin holding rank kk counted from the top -> S.sel_prefix = its index, S.sel_k = rank inside it, S.sel_cnt = its count
template <int Q, typename SH>
__device__ __forceinline__ void topk_find_bin(SH& S, int kk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int top = 256 * Q - 1 - (int)threadIdx.x * Q;  // thread t owns bins top, top - 1, .., top - Q + 1
  int v[Q], sum = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q) { v[q] = S.hist[top - q]; sum += v[q]; }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  if (lane == 63) S.wsum[wave] = inc;
  __syncthreads();
  for (int w = 0; w < wave; ++w) inc += S.wsum[w];
  int exc = inc - sum;
  if (exc < kk && kk <= inc) {  // exactly one thread
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (exc < kk && kk <= exc + v[q]) { S.sel_prefix = (uint32_t)(top - q); S.sel_k = kk - exc; S.sel_cnt = v[q]; }
      exc += v[q];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long long ldx, const int32_t* __restrict__ row_ids,
                                                         const int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_idx, int cols) {
  __shared__ TopkStreamShared S;
  const int u = row_ids[blockIdx.x];
  const int beg = out_ptr[u];
  int k = out_ptr[u + 1] - beg;
  if (k <= 0) return;  // uniform per block
  if (k > cols) k = cols;
  const float* __restrict__ r = x + (long long)blockIdx.x * ldx;
  const bool vec = (ldx % 4 == 0) && (((uintptr_t)x) & 15) == 0;
  const int c4 = vec ? cols / 4 : 0;  // columns [0, 4 c4) by 16-byte loads, the rest one by one
  // ---- read 1: the top 11 bits (four 16-byte loads in flight per thread: a 4 MB row is latency, not bandwidth, to one workgroup) ----
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256) S.hist[i] = 0;
  if (threadIdx.x == 0) S.n_cand = 0;
  __syncthreads();
  {
    int j = threadIdx.x;
    for (; j + 768 < c4; j += 1024) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const float4*>(r)[j + 256 * q];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        atomicAdd(&S.hist[fkey(v[q].x) >> 21], 1); atomicAdd(&S.hist[fkey(v[q].y) >> 21], 1);
        atomicAdd(&S.hist[fkey(v[q].z) >> 21], 1); atomicAdd(&S.hist[fkey(v[q].w) >> 21], 1);
      }
    }
    for (; j < c4; j += 256) {
      const float4 v = reinterpret_cast<const float4*>(r)[j];
      atomicAdd(&S.hist[fkey(v.x) >> 21], 1); atomicAdd(&S.hist[fkey(v.y) >> 21], 1);
      atomicAdd(&S.hist[fkey(v.z) >> 21], 1); atomicAdd(&S.hist[fkey(v.w) >> 21], 1);
    }
  }
  for (int j = 4 * c4 + threadIdx.x; j < cols; j += 256) atomicAdd(&S.hist[fkey(r[j]) >> 21], 1);
  __syncthreads();
  topk_find_bin<TOPK_BINS / 256>(S, k);
  const uint32_t top11 = S.sel_prefix;
  uint32_t prefix = top11 << 21, mask = 0xffe00000u;
  int kk = S.sel_k;
  const bool in_lds = S.sel_cnt <= TOPK_CAND;  // uniform
  // ---- read 2: the bucket's keys into LDS ----
  if (in_lds) {
    int j = threadIdx.x;
    for (; j + 768 < c4; j += 1024) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const float4*>(r)[j + 256 * q];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t ky[4] = {fkey(v[q].x), fkey(v[q].y), fkey(v[q].z), fkey(v[q].w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((ky[e] >> 21) == top11) S.cand[atomicAdd(&S.n_cand, 1)] = ky[e];
      }
    }
    for (; j < c4; j += 256) {
      const float4 v = reinterpret_cast<const float4*>(r)[j];
      const uint32_t ky[4] = {fkey(v.x), fkey(v.y), fkey(v.z), fkey(v.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((ky[e] >> 21) == top11) S.cand[atomicAdd(&S.n_cand, 1)] = ky[e];
    }
    for (int j = 4 * c4 + threadIdx.x; j < cols; j += 256) {
      const uint32_t key = fkey(r[j]);
      if ((key >> 21) == top11) S.cand[atomicAdd(&S.n_cand, 1)] = key;
    }
    __syncthreads();
  }
  // ---- the remaining 21 bits: digits of 8, 8 and 5 bits, over the collected keys (or, for an over-full bucket, over the row) ----
  const int n_cand = in_lds ? S.n_cand : 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 13 : (pass == 1 ? 5 : 0);
    const uint32_t dm = pass == 2 ? 31u : 255u;
    S.hist[threadIdx.x] = 0;
    __syncthreads();
    if (in_lds) {
      for (int i = threadIdx.x; i < n_cand; i += 256) {
        const uint32_t key = S.cand[i];
        if ((key & mask) == prefix) atomicAdd(&S.hist[(key >> shift) & dm], 1);
      }
    } else {
      for (int j = threadIdx.x; j < cols; j += 256) {
        const uint32_t key = fkey(r[j]);
        if ((key & mask) == prefix) atomicAdd(&S.hist[(key >> shift) & dm], 1);
      }
    }
    __syncthreads();
    topk_find_bin<1>(S, kk);
    prefix |= S.sel_prefix << shift;
    kk = S.sel_k;
    mask |= dm << shift;
  }
  const uint32_t T = prefix;  // the k-th largest key
  int need_ties = kk;         // its rank among the keys equal to T (>= 1): k - need_ties keys are larger -- no counting pass over the row
  int written = 0;            // uniform running output position
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- read 3: the ordered compaction, 2048 columns per iteration: a thread owns 8 consecutive columns (two 16-byte loads), the (above, tie) counts
  // of the threads are scanned inside the waves and across them through a double-buffered LDS record -- ONE barrier per 2048 columns (the first
  // version took 256 columns and two barriers per iteration: 3906 iterations of a 1M-column row, ~4 ms per row whatever the bandwidth) ----
  int buf = 0;
  for (int j0 = 0; j0 < cols; j0 += 2048, buf ^= 1) {
    const int base = j0 + (int)threadIdx.x * 8;
    uint32_t key[8];
    if (vec && base + 8 <= cols) {
      const float4 a = *reinterpret_cast<const float4*>(r + base), b = *reinterpret_cast<const float4*>(r + base + 4);
      key[0] = fkey(a.x); key[1] = fkey(a.y); key[2] = fkey(a.z); key[3] = fkey(a.w);
      key[4] = fkey(b.x); key[5] = fkey(b.y); key[6] = fkey(b.z); key[7] = fkey(b.w);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) key[e] = base + e < cols ? fkey(r[base + e]) : 0u;
    }
    unsigned gm = 0u, tm = 0u;  // this thread's columns above the threshold / equal to it
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = base + e < cols;
      gm |= (in && key[e] > T) ? (1u << e) : 0u;
      tm |= (in && key[e] == T) ? (1u << e) : 0u;
    }
    const int mine = __popc(gm) | (__popc(tm) << 16);  // both counts in one scan (at most 2048 each)
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int nb = __shfl_up(inc, o, 64);
      if (lane >= o) inc += nb;
    }
    if (lane == 63) S.cscan[buf][wave] = inc;
    __syncthreads();
    int before = inc - mine, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const int tw = S.cscan[buf][w2];
      if (w2 < wave) before += tw;
      total += tw;
    }
    const int gbase = before & 0xffff, tbase = before >> 16, gtot = total & 0xffff, ttot = total >> 16;
    // output order inside the chunk: ascending column => rank of an element = (# above before it) + (# TAKEN ties before it); ties are taken in column
    // order while the budget lasts
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned lower = (1u << e) - 1u;
      const int ties_before = tbase + __popc(tm & lower);
      const bool g = (gm >> e) & 1u, tk = ((tm >> e) & 1u) && ties_before < need_ties;
      if (g || tk) out_idx[beg + written + gbase + __popc(gm & lower) + min(ties_before, need_ties)] = base + e;
    }
    const int ties_taken_chunk = min(ttot, need_ties);
    written += gtot + ties_taken_chunk;
    need_ties -= ties_taken_chunk;
  }
}

// ---- register-resident selection for rows of up to 256 * NC columns (NC = 32: 8192; NC = 80: 20480) --------------------------------
// The row is read from memory ONCE (order-preserving keys held in registers through the four radix passes and the compaction), the
// bucket holding rank k is found by a parallel suffix scan over the 256 bins instead of a one-thread walk, and the compaction
// needs no barrier per 256-column chunk (per-(chunk, wave) counts go to LDS once, two waves scan them, every thread then knows the
// output positions of its elements).  Selection rule: the k largest; ties at the threshold -> smaller columns.
template <int NC>
struct TopkShared {
  int hist[256];
  int wsum[4];
  int sm4[4];
  uint32_t sel_prefix;
  int sel_k;
  int cg[NC * 4 + 64], ct[NC * 4 + 64];
};

// k-th largest key of the row (k >= 1, k <= number of valid columns)
template <int NC>
__device__ __forceinline__ uint32_t topk_select_threshold(const uint32_t (&key)[NC], int cols, int k, TopkShared<NC>& S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t prefix = 0, mask = 0;
  int kk = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    S.hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (i * 256 + (int)threadIdx.x < cols && (key[i] & mask) == prefix) atomicAdd(&S.hist[(key[i] >> shift) & 255], 1);
    __syncthreads();
    // thread t owns bin 255 - t: an inclusive scan over t is a suffix sum over the bins
    const int v = S.hist[255 - threadIdx.x];
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(inc, o, 64);
      if (lane >= o) inc += n;
    }
    if (lane == 63) S.wsum[wave] = inc;
    __syncthreads();
    for (int w = 0; w < wave; ++w) inc += S.wsum[w];
    const int exc = inc - v;
    if (exc < kk && kk <= inc) {  // exactly one bin: the one holding rank kk
      S.sel_prefix = prefix | ((uint32_t)(255 - threadIdx.x) << shift);
      S.sel_k = kk - exc;
    }
    __syncthreads();
    prefix = S.sel_prefix;
    kk = S.sel_k;
    mask |= 255u << shift;
  }
  return prefix;
}

// pos[i] = position (0 .. k-1, ascending column order) of this thread's i-th element among the selected ones, or -1
template <int NC>
__device__ __forceinline__ void topk_positions(const uint32_t (&key)[NC], int cols, int k, uint32_t T, TopkShared<NC>& S, int (&pos)[NC]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int c = 0;
#pragma unroll
  for (int i = 0; i < NC; ++i) c += (i * 256 + (int)threadIdx.x < cols && key[i] > T) ? 1 : 0;
  const int n_gt = block_sum_int(c, S.sm4);
  const int need_ties = k - n_gt;  // >= 1 by construction
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const bool valid = i * 256 + (int)threadIdx.x < cols;
    const unsigned long long mg = __ballot(valid && key[i] > T), mt = __ballot(valid && key[i] == T);
    if (lane == 0) {
      S.cg[i * 4 + wave] = __popcll(mg);
      S.ct[i * 4 + wave] = __popcll(mt);
    }
    pos[i] = (__popcll(mg & below) << 8) | __popcll(mt & below);  // ranks inside the wave's part of the chunk, for now
  }
  __syncthreads();
  if (wave < 2) {  // wave 0: exclusive prefix of the gt counts in (chunk, wave) order; wave 1: of the tie counts
    int* a = wave == 0 ? S.cg : S.ct;
    constexpr int PER = (NC * 4 + 63) / 64;
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = lane * PER + q;
      loc[q] = idx < NC * 4 ? a[idx] : 0;
      sum += loc[q];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(inc, o, 64);
      if (lane >= o) inc += n;
    }
    int run = inc - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = lane * PER + q;
      if (idx < NC * 4) a[idx] = run;
      run += loc[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const bool valid = i * 256 + (int)threadIdx.x < cols;
    const bool gt = valid && key[i] > T, tie = valid && key[i] == T;
    const int gt_before = S.cg[i * 4 + wave] + (pos[i] >> 8);
    const int tie_rank = S.ct[i * 4 + wave] + (pos[i] & 255);
    pos[i] = (gt || (tie && tie_rank < need_ties)) ? gt_before + min(tie_rank, need_ties) : -1;
  }
}

// k <= TOPK_SMALL (the light users of the TikTok degree law: 99 % of the rows, k = 1 .. 4): k rounds of a block-wide arg-max in the
// order (value descending, column ascending) -- exactly the set the radix selection picks (the k largest, ties at the threshold to the
// smaller columns) -- instead of four 256-bin radix passes.  The row is read once with 16-byte loads; a round is 2 x 28 compares per
// lane, a wave reduction and one LDS exchange.
constexpr int TOPK_SMALL = 8;
__device__ __forceinline__ unsigned long long topk_comb(float v, int col) {  // larger = better
  return ((unsigned long long)fkey(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)col);
}
template <int MAXQ>  // 16-byte pieces per lane: rows up to MAXQ * 1024 columns
__device__ __forceinline__ void topk_small_row(const float* __restrict__ r, int cols, int k, int32_t* __restrict__ out) {
  __shared__ unsigned long long best_w[2][4];  // (two sets: a round writes the one the previous round's readers are not using -- one barrier per round)
  __shared__ int picked[TOPK_SMALL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = (cols + 1023) / 1024;
  // (r4) the order-preserving keys are formed ONCE (they were re-derived in every round), and a round re-scans only in the wave that held the previous
  // winner: every other lane's best candidate is still below that winner and still its best (3.5 x fewer VALU instructions at the mean k of 5).
  uint32_t key[MAXQ][4];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int c = (q * 256 + (int)threadIdx.x) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nq) {
      if (c + 3 < cols) v = *reinterpret_cast<const float4*>(r + c);
      else {
        if (c < cols) v.x = r[c];
        if (c + 1 < cols) v.y = r[c + 1];
        if (c + 2 < cols) v.z = r[c + 2];
      }
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) key[q][j] = (q < nq && c + j < cols) ? fkey(e[j]) : 0u;  // 0: below every float's key, never picked (k <= cols)
  }
  const uint32_t lo0 = 0xffffffffu - (unsigned)threadIdx.x * 4u;  // low word of element (q, j): lo0 - (q * 1024 + j) = ~column
  auto scan = [&](unsigned long long prev) {  // the lane's best candidate strictly below `prev` in the order (value descending, column ascending)
    unsigned long long best = 0ull;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long cb = key[q][j] ? (((unsigned long long)key[q][j] << 32) | (unsigned long long)(lo0 - (unsigned)(q * 1024 + j))) : 0ull;
        if (cb < prev && cb > best) best = cb;
      }
    return best;
  };
  unsigned long long mine = scan(~0ull);
  for (int round = 0; round < k; ++round) {
    unsigned long long best = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) best_w[round & 1][wave] = best;
    __syncthreads();  // (the set written two rounds ago was read before the previous round's barrier)
    unsigned long long b = best_w[round & 1][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = best_w[round & 1][w] > b ? best_w[round & 1][w] : b;
    if (threadIdx.x == 0) picked[round] = (int)(0xffffffffu - (unsigned)(b & 0xffffffffull));
    if (best == b && round + 1 < k) mine = scan(b);  // wave-uniform: the wave that held the winner; its other lanes find their old candidate again
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // ascending column order, like the radix path's ordered compaction
    int c[TOPK_SMALL];
    for (int i = 0; i < k; ++i) c[i] = picked[i];
    for (int i = 1; i < k; ++i) {
      const int x = c[i];
      int j = i - 1;
      while (j >= 0 && c[j] > x) { c[j + 1] = c[j]; --j; }
      c[j + 1] = x;
    }
    for (int i = 0; i < k; ++i) out[i] = c[i];
  }
}

template <int NC>
__global__ __launch_bounds__(256) void topk_rows_reg_kernel(const float* __restrict__ x, long long ldx, const int32_t* __restrict__ row_ids,
                                                             const int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_idx, int cols) {
  __shared__ TopkShared<NC> S;
  const int u = row_ids[blockIdx.x];
  const int beg = out_ptr[u];
  int k = out_ptr[u + 1] - beg;
  if (k <= 0) return;  // uniform per block
  if (k > cols) k = cols;
  const float* __restrict__ r = x + (long long)blockIdx.x * ldx;
  if (k <= TOPK_SMALL && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0) {  // (uniform per block)
    topk_small_row<NC / 4>(r, cols, k, out_idx + beg);
    return;
  }
  uint32_t key[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + threadIdx.x;
    key[i] = j < cols ? fkey(r[j]) : 0u;
  }
  const uint32_t T = topk_select_threshold<NC>(key, cols, k, S);
  int pos[NC];
  topk_positions<NC>(key, cols, k, T, S, pos);
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (pos[i] >= 0) out_idx[beg + pos[i]] = i * 256 + threadIdx.x;
}

// Evaluation (Coach.testEpoch, Main.py:403-411), register-resident: the user's TRAIN items are set to -1e8 in the score row (in
// place, like the streaming kernel below), the k winners are found as above, parked in LDS and ordered by a rank sort
// (descending score, equal scores: smaller column first).  k <= 64.
template <int NC>
__global__ __launch_bounds__(256) void eval_topk_reg_kernel(float* __restrict__ score, long long ld, const int32_t* __restrict__ users,
                                                             const int64_t* __restrict__ trn_ptr, const int64_t* __restrict__ trn_col, int cols, int k,
                                                             int32_t* __restrict__ out) {
  __shared__ TopkShared<NC> S;
  __shared__ uint32_t wkey[64];
  __shared__ int wcol[64];
  const int u = users[blockIdx.x];
  float* __restrict__ r = score + (long long)blockIdx.x * ld;
  for (long long p = trn_ptr[u] + threadIdx.x; p < trn_ptr[u + 1]; p += 256) r[trn_col[p]] = -1e8f;
  __syncthreads();
  const int kq = k < cols ? k : cols;
  uint32_t key[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + threadIdx.x;
    key[i] = j < cols ? fkey(r[j]) : 0u;
  }
  const uint32_t T = topk_select_threshold<NC>(key, cols, kq, S);
  int pos[NC];
  topk_positions<NC>(key, cols, kq, T, S, pos);
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (pos[i] >= 0) {
      wkey[pos[i]] = key[i];
      wcol[pos[i]] = i * 256 + threadIdx.x;
    }
  __syncthreads();
  if ((int)threadIdx.x < kq) {  // winners sit in ascending column order: rank = how many beat me (larger key, or equal key at a smaller column)
    const uint32_t mk = wkey[threadIdx.x];
    int rank = 0;
    for (int b = 0; b < kq; ++b) rank += (wkey[b] > mk || (wkey[b] == mk && b < (int)threadIdx.x)) ? 1 : 0;
    out[(long long)blockIdx.x * k + rank] = wcol[threadIdx.x];
  }
}

// Evaluation (Coach.testEpoch, Main.py:403-411): scores of one user against all items with the user's TRAIN items pushed to
// -1e8 (the reference multiplies by (1 - mask) and subtracts 1e8 * mask), then the k best columns in descending score order
// (ties -> smaller column id).  One workgroup per user; k <= 64.  k selection rounds of a workgroup-wide arg-max would cost
// k passes; instead: radix-select the k-th key, compact the <= k winners into LDS, sort them there (k is tiny).
// The mask is applied to `score` in place (the caller's buffer is scratch for the next user batch anyway).
__global__ __launch_bounds__(256) void eval_topk_kernel(float* __restrict__ score, long long ld, const int32_t* __restrict__ users,
                                                         const int64_t* __restrict__ trn_ptr, const int64_t* __restrict__ trn_col, int cols, int k,
                                                         int32_t* __restrict__ out, int only_flagged) {
  if (only_flagged && out[(long long)blockIdx.x * k] != -1) return;  // (uniform) rows eval_topk_stream_kernel finished
  __shared__ int hist[256];
  __shared__ uint32_t sel_prefix;
  __shared__ int sel_k, n_out;
  __shared__ float wv[64];
  __shared__ int wi[64];
  const int u = users[blockIdx.x];
  float* __restrict__ r = score + (long long)blockIdx.x * ld;
  for (long long p = trn_ptr[u] + threadIdx.x; p < trn_ptr[u + 1]; p += 256) r[trn_col[p]] = -1e8f;
  __syncthreads();
  uint32_t prefix = 0, mask = 0;
  int kk = k < cols ? k : cols;
  const int kq = kk;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < cols; j += 256) {
      const uint32_t key = fkey(r[j]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0, b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= kk) break;
        acc += hist[b];
      }
      sel_prefix = prefix | ((uint32_t)b << shift);
      sel_k = kk - acc;
      n_out = 0;
    }
    __syncthreads();
    prefix = sel_prefix;
    kk = sel_k;
    mask |= 255u << shift;
  }
  const uint32_t T = prefix;
  // winners: everything above the threshold, plus the first `kk` ties in column order (single thread resolves the few ties)
  for (int j = threadIdx.x; j < cols; j += 256)
    if (fkey(r[j]) > T) {
      const int o = atomicAdd(&n_out, 1);
      wv[o] = r[j];
      wi[o] = j;
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    int o = n_out;
    for (int j = 0; j < cols && o < kq; ++j)
      if (fkey(r[j]) == T) {
        wv[o] = r[j];
        wi[o] = j;
        ++o;
      }
    // insertion sort: descending score, ascending column on ties
    for (int a = 1; a < kq; ++a) {
      const float v = wv[a];
      const int id = wi[a];
      int b = a - 1;
      while (b >= 0 && (wv[b] < v || (wv[b] == v && wi[b] > id))) {
        wv[b + 1] = wv[b];
        wi[b + 1] = wi[b];
        --b;
      }
      wv[b + 1] = v;
      wi[b + 1] = id;
    }
    for (int a = 0; a < kq; ++a) out[(long long)blockIdx.x * k + a] = wi[a];
  }
}

// The same for rows beyond the register-resident forms (> 20480 columns), in TWO reads of the row (see topk_rows_kernel): an 11-bit histogram finds
// the bucket holding rank k; the second read collects that bucket's (key, column) pairs in LDS and the < k entries above it as winners; threshold,
// ties (smaller column first) and the final order are resolved in LDS.  (eval_topk_kernel above -- four radix passes, a winners pass and a
// one-thread scan of the row for the ties, ~50k dependent loads per row at 100k columns -- stays as the path of an over-full bucket, and was
// 13 s per evaluation of 1M users x 100k items; this one: see DESIGN section 7.)
struct EvalStreamShared {
  int hist[TOPK_BINS];
  uint32_t ckey[TOPK_CAND];
  int ccol[TOPK_CAND];
  int wsum[4];
  uint32_t sel_prefix;
  int sel_k, sel_cnt, n_cand, n_out, n_tie;
  uint32_t wkey[64];
  int wcol[64];
};
__global__ __launch_bounds__(256) void eval_topk_stream_kernel(float* __restrict__ score, long long ld, const int32_t* __restrict__ users,
                                                                const int64_t* __restrict__ trn_ptr, const int64_t* __restrict__ trn_col, int cols, int k,
                                                                int32_t* __restrict__ out) {
  __shared__ EvalStreamShared S;
  const int u = users[blockIdx.x];
  float* __restrict__ r = score + (long long)blockIdx.x * ld;
  for (long long p = trn_ptr[u] + threadIdx.x; p < trn_ptr[u + 1]; p += 256) r[trn_col[p]] = -1e8f;
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256) S.hist[i] = 0;
  if (threadIdx.x == 0) { S.n_cand = 0; S.n_out = 0; S.n_tie = 0; }
  __syncthreads();
  const int kq = k < cols ? k : cols;
  const bool vec = (ld % 4 == 0) && (((uintptr_t)score) & 15) == 0;
  const int c4 = vec ? cols / 4 : 0;
  for (int j = threadIdx.x; j < c4; j += 256) {
    const float4 v = reinterpret_cast<const float4*>(r)[j];
    atomicAdd(&S.hist[fkey(v.x) >> 21], 1); atomicAdd(&S.hist[fkey(v.y) >> 21], 1);
    atomicAdd(&S.hist[fkey(v.z) >> 21], 1); atomicAdd(&S.hist[fkey(v.w) >> 21], 1);
  }
  for (int j = 4 * c4 + threadIdx.x; j < cols; j += 256) atomicAdd(&S.hist[fkey(r[j]) >> 21], 1);
  __syncthreads();
  topk_find_bin<TOPK_BINS / 256>(S, kq);
  const uint32_t top11 = S.sel_prefix;
  if (S.sel_cnt > TOPK_CAND) {  // (uniform) a row concentrated on one value: flagged (out[row, 0] = -1) for eval_topk_kernel, launched behind this one
    if (threadIdx.x == 0) out[(long long)blockIdx.x * k] = -1;
    return;
  }
  uint32_t prefix = top11 << 21, mask = 0xffe00000u;
  int kk = S.sel_k;
  auto take = [&](uint32_t key, int col) {
    const uint32_t b = key >> 21;
    if (b > top11) { const int o = atomicAdd(&S.n_out, 1); S.wkey[o] = key; S.wcol[o] = col; }  // fewer than k entries lie above the bucket
    else if (b == top11) { const int c = atomicAdd(&S.n_cand, 1); S.ckey[c] = key; S.ccol[c] = col; }
  };
  for (int j = threadIdx.x; j < c4; j += 256) {
    const float4 v = reinterpret_cast<const float4*>(r)[j];
    take(fkey(v.x), 4 * j); take(fkey(v.y), 4 * j + 1); take(fkey(v.z), 4 * j + 2); take(fkey(v.w), 4 * j + 3);
  }
  for (int j = 4 * c4 + threadIdx.x; j < cols; j += 256) take(fkey(r[j]), j);
  __syncthreads();
  const int n_cand = S.n_cand;
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 13 : (pass == 1 ? 5 : 0);
    const uint32_t dm = pass == 2 ? 31u : 255u;
    S.hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_cand; i += 256) {
      const uint32_t key = S.ckey[i];
      if ((key & mask) == prefix) atomicAdd(&S.hist[(key >> shift) & dm], 1);
    }
    __syncthreads();
    topk_find_bin<1>(S, kk);
    prefix |= S.sel_prefix << shift;
    kk = S.sel_k;
    mask |= dm << shift;
  }
  const uint32_t T = prefix;
  const int need_ties = kk;  // >= 1
  for (int i = threadIdx.x; i < n_cand; i += 256) {
    const uint32_t key = S.ckey[i];
    if (key > T) { const int o = atomicAdd(&S.n_out, 1); S.wkey[o] = key; S.wcol[o] = S.ccol[i]; }
    else if (key == T) atomicAdd(&S.n_tie, 1);
  }
  __syncthreads();
  const bool all_ties = S.n_tie == need_ties;  // the usual case: the threshold value occurs once
  for (int i = threadIdx.x; i < n_cand; i += 256) {
    if (S.ckey[i] != T) continue;
    bool in = all_ties;
    if (!in) {  // the need_ties smallest columns among the ties
      const int col = S.ccol[i];
      int rank = 0;
      for (int j = 0; j < n_cand; ++j) rank += (S.ckey[j] == T && S.ccol[j] < col) ? 1 : 0;
      in = rank < need_ties;
    }
    if (in) { const int o = atomicAdd(&S.n_out, 1); S.wkey[o] = T; S.wcol[o] = S.ccol[i]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < kq) {  // descending key, ties -> smaller column
    const uint32_t mk = S.wkey[threadIdx.x];
    const int mc = S.wcol[threadIdx.x];
    int rank = 0;
    for (int b = 0; b < kq; ++b) rank += (S.wkey[b] > mk || (S.wkey[b] == mk && S.wcol[b] < mc)) ? 1 : 0;
    out[(long long)blockIdx.x * k + rank] = mc;
  }
}

// Coach.calcRes (Main.py:422-448) for one user per wave: lane j holds the j-th recommendation; recall = hits / |truth|,
// ndcg = sum_hit 1/log2(j+2) / sum_{l < min(|truth|, k)} 1/log2(l+2), precision = hits / k.  fp64 like the reference's Python floats.
__global__ __launch_bounds__(256) void eval_metrics_kernel(const int32_t* __restrict__ top, const int32_t* __restrict__ users,
                                                            const int64_t* __restrict__ tst_ptr, const int64_t* __restrict__ tst_col, int rows, int k,
                                                            double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;  // whole waves leave together; no workgroup barrier below
  const int u = users[r];
  const long long b = tst_ptr[u], e = tst_ptr[u + 1];
  const int mine = lane < k ? top[(long long)r * k + lane] : -1;
  bool hit = false;
  for (long long p = b; p < e; ++p) hit |= (lane < k) && ((long long)mine == tst_col[p]);
  double dcg = hit ? 1.0 / log2((double)lane + 2.0) : 0.0;
  const long long nt = e - b;
  double ideal = (lane < k && lane < nt) ? 1.0 / log2((double)lane + 2.0) : 0.0;
  double nh = hit ? 1.0 : 0.0;
  for (int o = 32; o > 0; o >>= 1) {
    dcg += __shfl_xor(dcg, o);
    ideal += __shfl_xor(ideal, o);
    nh += __shfl_xor(nh, o);
  }
  if (lane == 0) {
    out[(long long)r * 3 + 0] = nt > 0 ? nh / (double)nt : 0.0;
    out[(long long)r * 3 + 1] = nt > 0 ? dcg / ideal : 0.0;
    out[(long long)r * 3 + 2] = nh / (double)k;
  }
}

}  // namespace dmm
using namespace dmm;

extern "C" int dmm_eval_metrics(const int32_t* top, const int32_t* users, const int64_t* tst_ptr, const int64_t* tst_col, int rows, int k, double* out,
                                void* stream) {
  DMM_CHECK_ARG(top && users && tst_ptr && tst_col && out && rows > 0 && k >= 1 && k <= 64, "dmm_eval_metrics: bad arguments (k <= 64)");
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, top, users, tst_ptr, tst_col, rows, k, out);
  DMM_LAUNCH_CHECK("dmm_eval_metrics");
  return 0;
}

extern "C" int dmm_eval_topk(float* score, int64_t ld, const int32_t* users, const int64_t* trn_ptr, const int64_t* trn_col, int rows, int cols,
                             int k, int32_t* out, void* stream) {
  DMM_CHECK_ARG(score && users && trn_ptr && trn_col && out && rows > 0 && cols > 0 && k >= 1 && k <= 64, "dmm_eval_topk: bad arguments (k <= 64)");
  if (cols <= 256 * 32) hipLaunchKernelGGL(eval_topk_reg_kernel<32>, dim3(rows), dim3(256), 0, (hipStream_t)stream, score, (long long)ld, users, trn_ptr, trn_col, cols, k, out);
  else if (cols <= 256 * 80) hipLaunchKernelGGL(eval_topk_reg_kernel<80>, dim3(rows), dim3(256), 0, (hipStream_t)stream, score, (long long)ld, users, trn_ptr, trn_col, cols, k, out);
  else {
    static const int legacy = [] { const char* e = getenv("DMM_EVAL_TOPK_LEGACY"); return e ? atoi(e) : 0; }();  // (tests: the multi-pass kernel alone)
    if (!legacy) hipLaunchKernelGGL(eval_topk_stream_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, score, (long long)ld, users, trn_ptr, trn_col, cols, k, out);
    hipLaunchKernelGGL(eval_topk_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, score, (long long)ld, users, trn_ptr, trn_col, cols, k, out, legacy ? 0 : 1);
  }
  DMM_LAUNCH_CHECK("dmm_eval_topk");
  return 0;
}

extern "C" int dmm_qsample_temb(const float* x0, int64_t ldx0, const float* noise, const int32_t* t, const float* sqrt_ac,
                                const float* sqrt_1mac, const float* w_emb, const float* b_emb, int te, float* xcat, int64_t ldx, int B, int I,
                                int draw_noise, uint64_t seed, uint64_t offset, float* eps_out, void* stream) {
  DMM_CHECK_ARG(t && w_emb && b_emb && xcat && B > 0 && I > 0, "dmm_qsample_temb: null operand");
  DMM_CHECK_ARG(te >= 2 && te <= MAX_TE && ldx >= I + te, "dmm_qsample_temb: te=%d must be 2..%d and ldx >= I+te", te, MAX_TE);
  DMM_CHECK_ARG(!(noise || draw_noise) || (sqrt_ac && sqrt_1mac), "dmm_qsample_temb: schedule tables missing");
  hipLaunchKernelGGL(qsample_temb_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x0, (long long)ldx0, noise, t, sqrt_ac, sqrt_1mac, w_emb,
                     b_emb, te, xcat, (long long)ldx, I, draw_noise, (unsigned long long)seed, (unsigned long long)offset, eps_out,
                     (const uint32_t*)nullptr, 0ll);
  DMM_LAUNCH_CHECK("dmm_qsample_temb");
  return 0;
}

extern "C" int dmm_qsample_temb_bits(const uint32_t* x0bits, int64_t ldb, const float* noise, const int32_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                                     const float* w_emb, const float* b_emb, int te, float* xcat, int64_t ldx, int B, int I, int draw_noise, uint64_t seed,
                                     uint64_t offset, float* eps_out, void* stream) {
  DMM_CHECK_ARG(x0bits && t && w_emb && b_emb && xcat && B > 0 && I > 0 && ldb * 32 >= I, "dmm_qsample_temb_bits: bad arguments");
  DMM_CHECK_ARG(te >= 2 && te <= MAX_TE && ldx >= I + te, "dmm_qsample_temb_bits: te=%d must be 2..%d and ldx >= I+te", te, MAX_TE);
  DMM_CHECK_ARG(!(noise || draw_noise) || (sqrt_ac && sqrt_1mac), "dmm_qsample_temb_bits: schedule tables missing");
  hipLaunchKernelGGL(qsample_temb_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, 0ll, noise, t, sqrt_ac, sqrt_1mac, w_emb,
                     b_emb, te, xcat, (long long)ldx, I, draw_noise, (unsigned long long)seed, (unsigned long long)offset, eps_out, x0bits, (long long)ldb);
  DMM_LAUNCH_CHECK("dmm_qsample_temb_bits");
  return 0;
}

// x_0 rows as bit masks: bits[b, w] bit k = 1 iff item 32 w + k is an interaction of user rows[b] (one workgroup per row: clear, then set)
__global__ __launch_bounds__(256) void csr_row_bits_kernel(const long long* __restrict__ ptr, const int32_t* __restrict__ col,
                                                           const long long* __restrict__ rows, uint32_t* __restrict__ bits, long long ldb, int I,
                                                           int col_lo) {
  extern __shared__ uint32_t wb[];
  const int b = blockIdx.x, W = (I + 31) >> 5;
  for (int w = threadIdx.x; w < W; w += 256) wb[w] = 0u;
  __syncthreads();
  const long long u = rows[b], beg = ptr[u], end = ptr[u + 1];
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int c = col[e] - col_lo;
    if ((unsigned)c < (unsigned)I) atomicOr(&wb[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  for (int w = threadIdx.x; w < ldb; w += 256) bits[(long long)b * ldb + w] = w < W ? wb[w] : 0u;
}

extern "C" int dmm_csr_row_bits_slice(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int col_lo, int I, uint32_t* bits, int64_t ldb,
                                      void* stream) {
  DMM_CHECK_ARG(ptr && col && rows && bits && B > 0 && I > 0 && col_lo >= 0 && ldb * 32 >= I, "dmm_csr_row_bits: bad arguments (ldb words >= ceil(I / 32))");
  const size_t lds = (size_t)((I + 31) >> 5) * 4;
  DMM_CHECK_ARG(lds <= 60 * 1024, "dmm_csr_row_bits: I=%d too large for the LDS row mask (<= 491520)", I);
  hipLaunchKernelGGL(csr_row_bits_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, bits,
                     (long long)ldb, I, col_lo);
  DMM_LAUNCH_CHECK("dmm_csr_row_bits");
  return 0;
}
extern "C" int dmm_csr_row_bits(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, uint32_t* bits, int64_t ldb, void* stream) {
  return dmm_csr_row_bits_slice(ptr, col, rows, B, 0, I, bits, ldb, stream);
}

extern "C" int dmm_csr_dense_rows_slice(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int col_lo, int I, float* out, int64_t ldo,
                                        void* stream) {
  DMM_CHECK_ARG(ptr && col && rows && out && B > 0 && I > 0 && col_lo >= 0 && ldo >= I && ldo % 4 == 0,
                "dmm_csr_dense_rows: bad arguments (ldo must be a multiple of 4 >= I)");
  hipLaunchKernelGGL(csr_dense_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, out, (long long)ldo, I,
                     col_lo);
  DMM_LAUNCH_CHECK("dmm_csr_dense_rows");
  return 0;
}
extern "C" int dmm_csr_dense_rows(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, float* out, int64_t ldo, void* stream) {
  return dmm_csr_dense_rows_slice(ptr, col, rows, B, 0, I, out, ldo, stream);
}
static int csr_gather_sum_launch(const char* who, const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const void* Wt, int64_t ldw,
                                 const float* bias, int H, int act, void* out16, int64_t ldo, float* z, int64_t ldz, void* stream) {
  DMM_CHECK_ARG(ptr && col && rows && Wt && bias && out16 && B > 0 && H > 0 && H % 4 == 0 && ldw >= H && ldw % 4 == 0 && ldo >= H && ldo % 4 == 0,
                "%s: bad arguments (H, ldw, ldo multiples of 4)", who);
  DMM_CHECK_ARG(!z || (ldz >= H && ldz % 4 == 0 && ((uintptr_t)z & 15) == 0), "%s: z must be 16-byte aligned with ldz >= H, ldz %% 4 == 0", who);
  DMM_CHECK_ARG(act == DMM_ACT_NONE || act == DMM_ACT_TANH, "%s: act must be NONE or TANH", who);
  CsrGatherBatch batch;
  memset(&batch, 0, sizeof(batch));
  batch.p[0] = CsrGatherP{(const __bf16*)Wt, (long long)ldw, bias, (__bf16*)out16, (long long)ldo, z, (long long)ldz};
  if (act == DMM_ACT_TANH)
    hipLaunchKernelGGL(csr_gather_sum_kernel<1>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, batch, H);
  else
    hipLaunchKernelGGL(csr_gather_sum_kernel<0>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, batch, H);
  DMM_LAUNCH_CHECK(who);
  return 0;
}
extern "C" int dmm_csr_gather_sum_latent_batched(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const dmm_csr_gather_problem* p, int n,
                                                 int H, void* stream) {
  DMM_CHECK_ARG(ptr && col && rows && p && n >= 1 && n <= 4 && B > 0 && H > 0 && H % 4 == 0, "dmm_csr_gather_sum_latent_batched: bad arguments (1..4 problems)");
  CsrGatherBatch batch;
  memset(&batch, 0, sizeof(batch));
  for (int i = 0; i < n; ++i) {
    DMM_CHECK_ARG(p[i].Wt && p[i].bias && p[i].out16 && p[i].z && p[i].ldw >= H && p[i].ldw % 4 == 0 && p[i].ldo >= H && p[i].ldo % 4 == 0 && p[i].ldz >= H &&
                      p[i].ldz % 4 == 0 && ((uintptr_t)p[i].z & 15) == 0,
                  "dmm_csr_gather_sum_latent_batched: problem %d: operands must be set, leading dimensions multiples of 4 >= H, z 16-byte aligned", i);
    batch.p[i] = CsrGatherP{(const __bf16*)p[i].Wt, (long long)p[i].ldw, p[i].bias, (__bf16*)p[i].out16, (long long)p[i].ldo, p[i].z, (long long)p[i].ldz};
  }
  hipLaunchKernelGGL(csr_gather_sum_kernel<1>, dim3(B, n), dim3(256), 0, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, batch, H);
  DMM_LAUNCH_CHECK("dmm_csr_gather_sum_latent_batched");
  return 0;
}
extern "C" int dmm_csr_gather_sum_bf16(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const void* Wt, int64_t ldw, const float* bias,
                                       int H, int act, void* out16, int64_t ldo, void* stream) {
  return csr_gather_sum_launch("dmm_csr_gather_sum_bf16", ptr, col, rows, B, Wt, ldw, bias, H, act, out16, ldo, nullptr, 0, stream);
}
extern "C" int dmm_csr_gather_sum_latent(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const void* Wt, int64_t ldw, const float* bias,
                                         int H, float* z, int64_t ldz, void* out16, int64_t ldo, void* stream) {
  DMM_CHECK_ARG(z, "dmm_csr_gather_sum_latent: z is NULL");
  return csr_gather_sum_launch("dmm_csr_gather_sum_latent", ptr, col, rows, B, Wt, ldw, bias, H, DMM_ACT_TANH, out16, ldo, z, ldz, stream);
}
extern "C" int dmm_csr_add_rows(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, float c, float* y, int64_t ldy, void* y16,
                                int64_t ldy16, void* stream) {
  DMM_CHECK_ARG(ptr && col && rows && (y || y16) && B > 0 && I > 0 && (!y || ldy >= I) && (!y16 || ldy16 >= I), "dmm_csr_add_rows: bad arguments");
  hipLaunchKernelGGL(csr_add_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)ptr, col, (const long long*)rows, c, y,
                     (long long)ldy, (__bf16*)y16, (long long)ldy16, I);
  DMM_LAUNCH_CHECK("dmm_csr_add_rows");
  return 0;
}

extern "C" int dmm_rows_axpby(const float* x, int64_t ldx, const float* y, int64_t ldy, const int32_t* t, const float* a_table, const float* c_table,
                              float* out, int64_t ldo, int B, int I, void* stream) {
  DMM_CHECK_ARG(x && y && t && a_table && c_table && out && B > 0 && I > 0 && ldx >= I && ldy >= I && ldo >= I, "dmm_rows_axpby: bad arguments");
  hipLaunchKernelGGL(rows_axpby_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, y, (long long)ldy, t, a_table, c_table, out,
                     (long long)ldo, I);
  DMM_LAUNCH_CHECK("dmm_rows_axpby");
  return 0;
}

extern "C" int dmm_noised_start(const float* x0, int64_t ldx0, const float* randn, float sqrt_ac, float sqrt_1mac, float* xT, int64_t ldx, int B, int I,
                                uint64_t seed, uint64_t offset, void* stream) {
  DMM_CHECK_ARG(x0 && xT && B > 0 && I > 0, "dmm_noised_start: bad arguments");
  hipLaunchKernelGGL(noised_start_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x0, (long long)ldx0, randn, sqrt_ac, sqrt_1mac, xT, (long long)ldx, I,
                     (unsigned long long)seed, (unsigned long long)offset);
  DMM_LAUNCH_CHECK("dmm_noised_start");
  return 0;
}

extern "C" int dmm_temb_bwd(const int32_t* t, const float* dxcat, int64_t ldx, int I, int te, int B, float* dw_emb, float* db_emb, void* stream) {
  DMM_CHECK_ARG(t && dxcat && dw_emb && db_emb && te >= 2 && te <= MAX_TE && B > 0, "dmm_temb_bwd: bad arguments");
  hipLaunchKernelGGL(temb_bwd_kernel, dim3(cdiv(B, TEMB_WG_ROWS)), dim3(1024), 0, (hipStream_t)stream, t, dxcat, (long long)ldx, I, te, B, dw_emb, db_emb);
  DMM_LAUNCH_CHECK("dmm_temb_bwd");
  return 0;
}

extern "C" int dmm_diff_loss(const float* out, const float* x0, int64_t ld, const double* w, const float* um, const float* ui, int B, int I, int d,
                             float sim_weight, double reg_term, double gscale, double* loss, float* dout, float* dum, const double* gscale_dev,
                             void* stream) {
  DMM_CHECK_ARG(out && x0 && w && um && ui && (loss || dum) && B > 0 && I > 0 && d > 0, "dmm_diff_loss: bad arguments");
  hipLaunchKernelGGL(diff_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, out, x0, (long long)ld, w, um, ui, I, d, sim_weight, reg_term,
                     gscale, loss, dout, dum, (const uint32_t*)nullptr, 0ll, gscale_dev);
  DMM_LAUNCH_CHECK("dmm_diff_loss");
  return 0;
}

extern "C" int dmm_diff_loss_bits(const float* out, int64_t ld, const uint32_t* x0bits, int64_t ldb, const double* w, const float* um, const float* ui, int B,
                                  int I, int d, float sim_weight, double reg_term, double* loss, void* stream) {
  DMM_CHECK_ARG(out && x0bits && w && um && ui && loss && B > 0 && I > 0 && d > 0 && ldb * 32 >= I, "dmm_diff_loss_bits: bad arguments");
  hipLaunchKernelGGL(diff_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, out, (const float*)nullptr, (long long)ld, w, um, ui, I, d, sim_weight,
                     reg_term, 0.0, loss, (float*)nullptr, (float*)nullptr, x0bits, (long long)ldb, (const double*)nullptr);
  DMM_LAUNCH_CHECK("dmm_diff_loss_bits");
  return 0;
}

extern "C" int dmm_topk_rows(const float* x, int64_t ldx, const int32_t* row_ids, const int32_t* out_ptr, int32_t* out_idx, int rows, int cols,
                             void* stream) {
  DMM_CHECK_ARG(x && row_ids && out_ptr && out_idx && rows > 0 && cols > 0, "dmm_topk_rows: bad arguments");
  if (cols <= 256 * 32) hipLaunchKernelGGL(topk_rows_reg_kernel<32>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, row_ids, out_ptr, out_idx, cols);
  else if (cols <= 256 * 80) hipLaunchKernelGGL(topk_rows_reg_kernel<80>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, row_ids, out_ptr, out_idx, cols);
  else hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, row_ids, out_ptr, out_idx, cols);
  DMM_LAUNCH_CHECK("dmm_topk_rows");
  return 0;
}
