// Memory-bound pieces of the joint step: row normalisation, modal mixing, reductions, fused Adam.
// All are HBM/L2-bandwidth kernels: 16-byte accesses per lane, grid-stride, <= 2048 workgroups.
#include <stdarg.h>

#include <string.h>

#include <algorithm>

#include "common.h"

namespace dmm {
// last-error text of the CALLING thread (errno-style): the library keeps no mutable state shared between host threads
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static inline unsigned grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace dmm

extern "C" int dmm_version(void) { return DMM_VERSION; }
extern "C" const char* dmm_last_error_string(void) { return dmm::g_err; }

namespace dmm {

// ---------------------------------------------------------------------------------------------
// row normalise: one group of LANES = D/4 lanes per row (same shape as the SpMM groups)
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           float* __restrict__ norm, int rows, __bf16* __restrict__ y16) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int l = threadIdx.x % LANES;
  for (int r = blockIdx.x * G + threadIdx.x / LANES; r < rows; r += gridDim.x * G) {  // group-uniform trip count
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * D + 4 * l);
    float ss = group_sum<LANES>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float nrm = sqrtf(ss);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    const float4 o = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    *reinterpret_cast<float4*>(y + (size_t)r * D + 4 * l) = o;
    if (y16) {  // the bf16 gather copy of the normalised rows, written by their producer
      __bf16 h[4] = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
      *reinterpret_cast<uint2*>(y16 + (size_t)r * D + 4 * l) = *reinterpret_cast<const uint2*>(h);
    }
    if (l == 0 && norm) norm[r] = nrm;
  }
}

template <int D>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ norm, float* __restrict__ dx, int rows) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int l = threadIdx.x % LANES;
  for (int r = blockIdx.x * G + threadIdx.x / LANES; r < rows; r += gridDim.x * G) {
    const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * D + 4 * l);
    const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * D + 4 * l);
    const float nrm = norm[r];
    float4 o;
    if (nrm < 1e-12f) {  // clamped branch of F.normalize: y = x / eps
      o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    } else {
      const float dot = group_sum<LANES>(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w);
      const float inv = 1.0f / nrm;
      o = make_float4((g.x - v.x * dot) * inv, (g.y - v.y * dot) * inv, (g.z - v.z * dot) * inv, (g.w - v.w * dot) * inv);
    }
    *reinterpret_cast<float4*>(dx + (size_t)r * D + 4 * l) = o;
  }
}

// Backward of F.normalize for up to 4 tensors in one launch (blockIdx.y), with the column sums of the result added to csum[] (the bias
// gradient of the Linear layer in front of the normalisation, Model.py:84-89): in the joint step this replaces 3 rownorm_bwd + 3 colsum launches
// over the modalities' [I, d] gradients.  Column sums: per-thread partials over the rows it handles, combined through LDS, one atomic
// per column and workgroup (csum pre-zeroed by the caller).
struct RownormBwdP { const float* dy; const float* y; const float* norm; float* dx; float* csum; int rows; };
struct RownormBwdBatch { RownormBwdP p[4]; };

template <int D>
__global__ __launch_bounds__(256) void rownorm_bwd_batched_kernel(const RownormBwdBatch batch) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  __shared__ float4 part[256];
  const RownormBwdP& P = batch.p[blockIdx.y];
  const float* __restrict__ dy = P.dy; const float* __restrict__ y = P.y; const float* __restrict__ norm = P.norm;
  float* __restrict__ dx = P.dx;
  const int rows = P.rows;
  const int l = threadIdx.x % LANES;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = blockIdx.x * G + threadIdx.x / LANES; r < rows; r += gridDim.x * G) {
    const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * D + 4 * l);
    const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * D + 4 * l);
    const float nrm = norm[r];
    float4 o;
    if (nrm < 1e-12f) {  // clamped branch of F.normalize: y = x / eps
      o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    } else {
      const float dot = group_sum<LANES>(g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w);
      const float inv = 1.0f / nrm;
      o = make_float4((g.x - v.x * dot) * inv, (g.y - v.y * dot) * inv, (g.z - v.z * dot) * inv, (g.w - v.w * dot) * inv);
    }
    *reinterpret_cast<float4*>(dx + (size_t)r * D + 4 * l) = o;
    cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
  }
  if (!P.csum) return;  // (uniform over the workgroup)
  part[threadIdx.x] = cs;
  __syncthreads();
  if (threadIdx.x < LANES) {
    float4 a = part[threadIdx.x];
    for (int g2 = 1; g2 < G; ++g2) {
      const float4 o = part[g2 * LANES + threadIdx.x];
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    float* c = P.csum + 4 * threadIdx.x;
    atomicAdd(c + 0, a.x); atomicAdd(c + 1, a.y); atomicAdd(c + 2, a.z); atomicAdd(c + 3, a.w);
  }
}

// ---------------------------------------------------------------------------------------------
// modal mixing
// ---------------------------------------------------------------------------------------------
struct MixPtrs {
  const float* z[3];
  float* dz[3];
  __bf16* dz16[3];  // optional bf16 copies of the updated dz (gather operands of the next SpMM)
};
typedef __bf16 ew_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_bf16x4(__bf16* base, int64_t i4, const float4& v) {
  reinterpret_cast<ew_bf16x4*>(base)[i4] = ew_bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

__device__ __forceinline__ void softmax3(const float* logits, int M, float (&w)[3]) {
  float mx = logits[0];
  for (int m = 1; m < M; ++m) mx = fmaxf(mx, logits[m]);
  float s = 0.f;
  for (int m = 0; m < 3; ++m) {
    w[m] = m < M ? expf(logits[m] - mx) : 0.f;
    s += w[m];
  }
  for (int m = 0; m < 3; ++m) w[m] /= s;
}

template <bool X16>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ bx, MixPtrs P, const float* __restrict__ logits,
                                                       int M, float maw, float* __restrict__ mx, __bf16* __restrict__ mx16, int64_t n4) {
  float w[3];
  softmax3(logits, M, w);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 b = reinterpret_cast<const float4*>(bx)[i];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {  // same association as the reference: w_m * (bx + maw * z_m), summed over m
      const float4 z = reinterpret_cast<const float4*>(P.z[m])[i];
      o.x += w[m] * (b.x + maw * z.x);
      o.y += w[m] * (b.y + maw * z.y);
      o.z += w[m] * (b.z + maw * z.z);
      o.w += w[m] * (b.w + maw * z.w);
    }
    reinterpret_cast<float4*>(mx)[i] = o;
    if (X16) st_bf16x4(mx16, i, o);
  }
}

// dbx = sw*dmx + extra ; dz_m += maw*w_m*dmx ; partial[blk][m] = sum dmx * (bx + maw z_m)
template <bool X16, bool OVER = false>  // OVER: dz_m = maw w_m dmx (dz is not read: nothing was accumulated into it before this launch)
__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ dmx, const float* __restrict__ bx, MixPtrs P,
                                                       const float* __restrict__ extra, float* __restrict__ dbx, __bf16* __restrict__ dbx16,
                                                       const float* __restrict__ logits, int M, float maw,
                                                       float* __restrict__ partials, int64_t n4, int64_t cnt0_end, int64_t cnt1_begin,
                                                       int64_t cnt1_end) {
  __shared__ float sm4[4];
  float w[3];
  softmax3(logits, M, w);
  const float sw = w[0] + w[1] + w[2];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(dmx)[i];
    const float4 b = reinterpret_cast<const float4*>(bx)[i];
    float4 o = make_float4(sw * g.x, sw * g.y, sw * g.z, sw * g.w);
    if (extra) {
      const float4 e = reinterpret_cast<const float4*>(extra)[i];
      o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
    }
    reinterpret_cast<float4*>(dbx)[i] = o;
    if (X16) st_bf16x4(dbx16, i, o);
    for (int m = 0; m < M; ++m) {
      const float4 z = reinterpret_cast<const float4*>(P.z[m])[i];
      // user-sharded layout: a replica counts its own users and its slice of the replicated item rows towards d(logits)
      if (i < cnt0_end || (i >= cnt1_begin && i < cnt1_end))
        acc[m] += g.x * (b.x + maw * z.x) + g.y * (b.y + maw * z.y) + g.z * (b.z + maw * z.z) + g.w * (b.w + maw * z.w);
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!OVER) d = reinterpret_cast<float4*>(P.dz[m])[i];
      const float c = maw * w[m];
      d.x += c * g.x; d.y += c * g.y; d.z += c * g.z; d.w += c * g.w;
      reinterpret_cast<float4*>(P.dz[m])[i] = d;
      if (X16) st_bf16x4(P.dz16[m], i, d);
    }
  }
  for (int m = 0; m < 3; ++m) {
    const float s = block_sum_256(acc[m], sm4);
    if (threadIdx.x == 0) partials[blockIdx.x * 3 + m] = s;
  }
}

__global__ __launch_bounds__(256) void mix_bwd_final_kernel(const float* __restrict__ partials, int nblk,
                                                             const float* __restrict__ logits, int M, float* __restrict__ dlogits) {
  __shared__ double sm4[4];
  double g[3];
  for (int m = 0; m < 3; ++m) {
    double a = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) a += (double)partials[i * 3 + m];
    g[m] = block_sum_256d(a, sm4);
  }
  if (threadIdx.x == 0) {
    float w[3];
    softmax3(logits, M, w);
    double dotwg = 0.0;
    for (int m = 0; m < M; ++m) dotwg += (double)w[m] * g[m];
    for (int m = 0; m < M; ++m) dlogits[m] = (float)((double)w[m] * (g[m] - dotwg));
  }
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ in, int64_t len, float scale,
                                                           float* __restrict__ out) {
  __shared__ double sm4[4];
  const float* p = in + (size_t)blockIdx.x * len;
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < len; i += 256) a += (double)p[i];
  a = block_sum_256d(a, sm4);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(a * (double)scale);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float sm4[4];
  float a = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    a += v * v;
  }
  a = block_sum_256(a, sm4);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

// The joint step's loss reductions in ONE launch (they were three: dmm_segment_sum, dmm_sumsq's two): blocks [0, n_seg) sum the row-loss segments
// (fp64, fixed order), the other blocks form partial sums of squares of x and the LAST of them to arrive adds the partials in index order -- the same
// values, the same order as the separate launches, so the result does not depend on which block is last.  `acc` (optional, fp64): the epoch's
// running sums, acc[k] += out[k] by the one thread that writes out[k] (Coach.trainEpoch added loss_parts after every step with a torch launch).
// counter: one int, zero before the first launch; the last block resets it.
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ seg_in, int n_seg, int64_t len, float seg_scale, float* __restrict__ seg_out,
                                                          const float* __restrict__ x, int64_t n, float x_scale, float* __restrict__ partial,
                                                          int* __restrict__ counter, float* __restrict__ x_out, double* __restrict__ acc, int acc_x) {
  __shared__ double smd[4];
  __shared__ float smf[4];
  __shared__ int last;
  if ((int)blockIdx.x < n_seg) {
    const float* p = seg_in + (size_t)blockIdx.x * len;
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < len; i += 256) a += (double)p[i];
    a = block_sum_256d(a, smd);
    if (threadIdx.x == 0) {
      const float o = (float)(a * (double)seg_scale);
      seg_out[blockIdx.x] = o;
      if (acc) acc[blockIdx.x] += (double)o;
    }
    return;
  }
  const int bx = (int)blockIdx.x - n_seg, gx = (int)gridDim.x - n_seg;
  float a = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n4; i += (int64_t)gx * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (bx == 0 && threadIdx.x < (n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    a += v * v;
  }
  a = block_sum_256(a, smf);
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + bx, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(counter, 1) == gx - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double t = 0.0;
  for (int i = threadIdx.x; i < gx; i += 256) t += (double)__hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum_256d(t, smd);
  if (threadIdx.x == 0) {
    const float o = (float)(t * (double)x_scale);
    x_out[0] = o;
    if (acc) acc[acc_x] += (double)o;
    *counter = 0;
  }
}

// up to 4 regions cleared by one launch (the joint step cleared its accumulation arena and the projections' weight gradients with two memsets)
struct ZeroRegions { float4* p[4]; long long n16[4]; int cnt; };
__global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroRegions Z) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < Z.cnt; ++r) {
    float4* __restrict__ p = Z.p[r];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < Z.n16[r]; i += (long long)gridDim.x * 256) p[i] = z;
  }
}

// ---------------------------------------------------------------------------------------------
// Adam
// ---------------------------------------------------------------------------------------------
struct AdamExtra {
  const float* g[4];
  int64_t n[4];
  int cnt;
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, AdamExtra ex, float l2c, float step_size,
                                                    float b1, float b2, float inv_sqrt_c2, float eps, int64_t l2_n, __bf16* __restrict__ p16,
                                                    int64_t p16_n, int64_t p16_lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pv = p[i];
    float gr = g[i];
    for (int k = 0; k < ex.cnt; ++k)
      if (i < ex.n[k]) gr += ex.g[k][i];
    if (i < l2_n) gr = fmaf(l2c, pv, gr);  // the l2 term covers a prefix (the embedding rows of a flat buffer that also holds other weights)
    const float mm = b1 * m[i] + (1.0f - b1) * gr;
    const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mm;
    v[i] = vv;
    const float pn = pv - step_size * mm / (sqrtf(vv) * inv_sqrt_c2 + eps);
    p[i] = pn;
    // the bf16 copy of a window of the parameters (the embedding table's gather copy; a Denoise's W1 | W2 GEMM operands), written by the step that
    // changes them
    if (i >= p16_lo && i < p16_lo + p16_n) p16[i - p16_lo] = (__bf16)pn;
  }
}

// ---------------------------------------------------------------------------------------------
// small elementwise ops of the diffusion path
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = a[i] * b[i];
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ proj,
                                                        const float* __restrict__ gate, float* __restrict__ dproj,
                                                        float* __restrict__ dpre, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = dout[i], s = gate[i];
    dproj[i] = d * s;
    dpre[i] = d * proj[i] * s * (1.0f - s);
  }
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                        float* __restrict__ dpre, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float t = h[i];
    dpre[i] = dh[i] * (1.0f - t * t);
  }
}
// x = tanh(x + bias[col]) in place (fp32, contiguous [rows, cols]) + a bf16 copy: layer 1's activation once its item-sliced partial products have been
// summed over the replicas (item-sharded Denoise); 4 columns per thread
__global__ __launch_bounds__(256) void bias_tanh_rows_kernel(float* __restrict__ x, const float* __restrict__ bias, int cols, int64_t n4,
                                                              __bf16* __restrict__ out16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    const int c = (int)((i * 4) % cols);
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    v.x = tanhf(v.x + b.x); v.y = tanhf(v.y + b.y); v.z = tanhf(v.z + b.z); v.w = tanhf(v.w + b.w);
    reinterpret_cast<float4*>(x)[i] = v;
    if (out16) {
      __bf16 o[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      reinterpret_cast<uint2*>(out16)[i] = *reinterpret_cast<const uint2*>(o);
    }
  }
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = fmaf(alpha, x[i], y[i]);
}
// column sums: grid (column groups of 64, row chunks); every wave strides its chunk's rows reading 256-byte row segments,
// the workgroup combines its 4 waves through LDS and adds the chunk's partial into out[] (fp32 atomic; out pre-zeroed)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int rows, int cols, int rows_per_chunk,
                                                      float* __restrict__ out) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float a = 0.f;
  if (c < cols)
    for (int r = r0 + w; r < r1; r += 4) a += x[(size_t)r * ldx + c];
  sm[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && c < cols) atomicAdd(out + c, sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

template <int D>
static int rownorm_launch(bool fwd, const float* a, const float* b, const float* nrm_in, float* out, float* nrm_out, int rows, hipStream_t s, void* out16) {
  constexpr int G = 1024 / D;
  const unsigned grid = grid_for(rows, G);
  if (fwd)
    hipLaunchKernelGGL(rownorm_fwd_kernel<D>, dim3(grid), dim3(256), 0, s, a, out, nrm_out, rows, (__bf16*)out16);
  else
    hipLaunchKernelGGL(rownorm_bwd_kernel<D>, dim3(grid), dim3(256), 0, s, a, b, nrm_in, out, rows);
  DMM_LAUNCH_CHECK("dmm_rownorm");
  return 0;
}
static int rownorm_dispatch(bool fwd, const float* a, const float* b, const float* nrm_in, float* out, float* nrm_out, int rows, int D, hipStream_t s,
                            void* out16 = nullptr) {
  switch (D) {
    case 8: return rownorm_launch<8>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
    case 16: return rownorm_launch<16>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
    case 32: return rownorm_launch<32>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
    case 64: return rownorm_launch<64>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
    case 128: return rownorm_launch<128>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
    case 256: return rownorm_launch<256>(fwd, a, b, nrm_in, out, nrm_out, rows, s, out16);
  }
  set_error("dmm_rownorm: D=%d unsupported (8,16,32,64,128,256)", D);
  return DMM_ERR_INVALID_ARG;
}
}  // namespace dmm

using namespace dmm;

extern "C" int dmm_rownorm_bwd_batched(const dmm_rownorm_bwd_problem* p, int n, int D, void* stream) {
  DMM_CHECK_ARG(p && n >= 1 && n <= 4, "dmm_rownorm_bwd_batched: 1..4 problems per launch");
  RownormBwdBatch b;
  memset(&b, 0, sizeof(b));
  int max_rows = 0;
  for (int i = 0; i < n; ++i) {
    DMM_CHECK_ARG(p[i].dy && p[i].y && p[i].norm && p[i].dx && p[i].rows >= 0, "dmm_rownorm_bwd_batched: null operand (problem %d)", i);
    b.p[i] = RownormBwdP{p[i].dy, p[i].y, p[i].norm, p[i].dx, p[i].csum, p[i].rows};
    max_rows = max_rows > p[i].rows ? max_rows : p[i].rows;
  }
  if (max_rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
#define RNB_CASE(DD)                                                                                                          \
  case DD: {                                                                                                                  \
    const unsigned gx = grid_for(max_rows, 1024 / DD);                                                                        \
    hipLaunchKernelGGL(rownorm_bwd_batched_kernel<DD>, dim3(gx > 64 ? 64 : gx, n), dim3(256), 0, s, b); /* <= 64 atomics per column */ \
    break;                                                                                                                    \
  }
  switch (D) {
    RNB_CASE(8) RNB_CASE(16) RNB_CASE(32) RNB_CASE(64) RNB_CASE(128) RNB_CASE(256)
    default:
      set_error("dmm_rownorm_bwd_batched: D=%d unsupported (8,16,32,64,128,256)", D);
      return DMM_ERR_INVALID_ARG;
  }
#undef RNB_CASE
  DMM_LAUNCH_CHECK("dmm_rownorm_bwd_batched");
  return 0;
}

extern "C" int dmm_rownorm_fwd(const float* x, float* y, float* norm, int rows, int D, void* stream) {
  DMM_CHECK_ARG(x && y && rows >= 0, "dmm_rownorm_fwd: null operand");
  if (rows == 0) return 0;
  return rownorm_dispatch(true, x, nullptr, nullptr, y, norm, rows, D, (hipStream_t)stream);
}
extern "C" int dmm_rownorm_fwd_x16(const float* x, float* y, void* y16, float* norm, int rows, int D, void* stream) {
  DMM_CHECK_ARG(x && y && y16 && rows >= 0 && ((uintptr_t)y16 & 7) == 0, "dmm_rownorm_fwd_x16: null or misaligned operand");
  if (rows == 0) return 0;
  return rownorm_dispatch(true, x, nullptr, nullptr, y, norm, rows, D, (hipStream_t)stream, y16);
}
extern "C" int dmm_rownorm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D, void* stream) {
  DMM_CHECK_ARG(dy && y && norm && dx && rows >= 0, "dmm_rownorm_bwd: null operand");
  if (rows == 0) return 0;
  return rownorm_dispatch(false, dy, y, norm, dx, nullptr, rows, D, (hipStream_t)stream);
}

static int mix_fwd_launch(const float* bx, const float* const* z, const float* w_logits, int M, float maw, float* mx, void* mx16, int64_t n, void* stream) {
  DMM_CHECK_ARG(bx && z && w_logits && mx && (M == 2 || M == 3) && n % 4 == 0, "dmm_gcn_mix_fwd: bad arguments (M=%d n=%lld)", M, (long long)n);
  MixPtrs P = {};
  for (int m = 0; m < M; ++m) P.z[m] = z[m];
  if (mx16) hipLaunchKernelGGL(mix_fwd_kernel<true>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, bx, P, w_logits, M, maw, mx, (__bf16*)mx16, n / 4);
  else hipLaunchKernelGGL(mix_fwd_kernel<false>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, bx, P, w_logits, M, maw, mx, (__bf16*)nullptr, n / 4);
  DMM_LAUNCH_CHECK("dmm_gcn_mix_fwd");
  return 0;
}
extern "C" int dmm_gcn_mix_fwd(const float* bx, const float* const* z, const float* w_logits, int M, float maw, float* mx,
                               int64_t n, void* stream) {
  return mix_fwd_launch(bx, z, w_logits, M, maw, mx, nullptr, n, stream);
}
extern "C" int dmm_gcn_mix_fwd_x16(const float* bx, const float* const* z, const float* w_logits, int M, float maw, float* mx, void* mx16,
                                   int64_t n, void* stream) {
  DMM_CHECK_ARG(mx16 && ((uintptr_t)mx16 & 7) == 0, "dmm_gcn_mix_fwd_x16: mx16 must be an 8-byte aligned buffer");
  return mix_fwd_launch(bx, z, w_logits, M, maw, mx, mx16, n, stream);
}
extern "C" int64_t dmm_gcn_mix_partials(int64_t n) { return (int64_t)grid_for(n / 4, 256) * 3; }
static int mix_bwd_launch(const float* dmx, const float* bx, const float* const* z, float* const* dz, const float* dbx_extra, float* dbx,
                          const float* w_logits, float* dw_logits, int M, float maw, float* partials, int64_t n, int64_t count0_end, int64_t count1_begin,
                          int64_t count1_end, void* dbx16, void* const* dz16, void* stream, int dz_overwrite = 0) {
  DMM_CHECK_ARG(dmx && bx && z && dz && dbx && w_logits && dw_logits && partials && (M == 2 || M == 3) && n % 4 == 0,
                "dmm_gcn_mix_bwd: bad arguments");
  DMM_CHECK_ARG(!dz_overwrite || dz16, "dmm_gcn_mix_bwd: dz_overwrite is a mode of the x16 entry point");
  DMM_CHECK_ARG(count0_end % 4 == 0 && count1_begin % 4 == 0 && count1_end % 4 == 0 && 0 <= count0_end && count0_end <= n && count1_begin <= count1_end && count1_end <= n,
                "dmm_gcn_mix_bwd_counted: counted ranges must be multiples of 4 inside [0, n]");
  MixPtrs P = {};
  for (int m = 0; m < M; ++m) {
    P.z[m] = z[m];
    P.dz[m] = dz[m];
    if (dz16) {
      DMM_CHECK_ARG(dz16[m] && ((uintptr_t)dz16[m] & 7) == 0, "dmm_gcn_mix_bwd_x16: dz16[%d] must be an 8-byte aligned buffer", m);
      P.dz16[m] = (__bf16*)dz16[m];
    }
  }
  const unsigned grid = grid_for(n / 4, 256);
  if (dz16 && dz_overwrite) hipLaunchKernelGGL((mix_bwd_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dmx, bx, P, dbx_extra, dbx, (__bf16*)dbx16, w_logits, M,
                                               maw, partials, n / 4, count0_end / 4, count1_begin / 4, count1_end / 4);
  else if (dz16) hipLaunchKernelGGL(mix_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dmx, bx, P, dbx_extra, dbx, (__bf16*)dbx16, w_logits, M, maw, partials, n / 4,
                               count0_end / 4, count1_begin / 4, count1_end / 4);
  else hipLaunchKernelGGL(mix_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dmx, bx, P, dbx_extra, dbx, (__bf16*)nullptr, w_logits, M, maw, partials, n / 4,
                          count0_end / 4, count1_begin / 4, count1_end / 4);
  hipLaunchKernelGGL(mix_bwd_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, (int)grid, w_logits, M, dw_logits);
  DMM_LAUNCH_CHECK("dmm_gcn_mix_bwd");
  return 0;
}
extern "C" int dmm_gcn_mix_bwd(const float* dmx, const float* bx, const float* const* z, float* const* dz, const float* dbx_extra,
                               float* dbx, const float* w_logits, float* dw_logits, int M, float maw, float* partials, int64_t n,
                               void* stream) {
  return mix_bwd_launch(dmx, bx, z, dz, dbx_extra, dbx, w_logits, dw_logits, M, maw, partials, n, n, n, n, nullptr, nullptr, stream);
}
extern "C" int dmm_gcn_mix_bwd_counted(const float* dmx, const float* bx, const float* const* z, float* const* dz, const float* dbx_extra,
                                       float* dbx, const float* w_logits, float* dw_logits, int M, float maw, float* partials, int64_t n,
                                       int64_t count0_end, int64_t count1_begin, int64_t count1_end, void* stream) {
  return mix_bwd_launch(dmx, bx, z, dz, dbx_extra, dbx, w_logits, dw_logits, M, maw, partials, n, count0_end, count1_begin, count1_end, nullptr, nullptr, stream);
}
extern "C" int dmm_gcn_mix_bwd_x16(const float* dmx, const float* bx, const float* const* z, float* const* dz, const float* dbx_extra, float* dbx,
                                   const float* w_logits, float* dw_logits, int M, float maw, float* partials, int64_t n, int64_t count0_end,
                                   int64_t count1_begin, int64_t count1_end, void* dbx16, void* const* dz16, int dz_overwrite, void* stream) {
  DMM_CHECK_ARG(dbx16 && dz16 && ((uintptr_t)dbx16 & 7) == 0, "dmm_gcn_mix_bwd_x16: dbx16 / dz16 missing or misaligned");
  return mix_bwd_launch(dmx, bx, z, dz, dbx_extra, dbx, w_logits, dw_logits, M, maw, partials, n, count0_end, count1_begin, count1_end, dbx16, dz16, stream, dz_overwrite);
}

// The rows a joint minibatch's scatter-adds touch -- users[b], U + pos[b], U + neg[b] -- of up to 4 [N, D] tensors: dst row = bf16(src row), or
// zeros where src[t] is NULL.  (The gradient tensors the losses scatter into are zero everywhere else: their bf16 gather copies are kept by
// casting just these rows and clearing them again after use, instead of a pass over all N rows.)
struct BatchRowsP { const float* src[4]; __bf16* dst[4]; };
__global__ __launch_bounds__(256) void cast_batch_rows_kernel(BatchRowsP P, int D, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
                                                              const int32_t* __restrict__ neg, int B, int U, const int32_t* __restrict__ seg) {
  const int lanes = D / 8, per = 256 / lanes;
  const int l = threadIdx.x % lanes;
  const float* __restrict__ src = P.src[blockIdx.y];
  __bf16* __restrict__ dst = P.dst[blockIdx.y];
  const int s0 = seg ? seg[0] : 0, s1 = seg ? seg[1] : B;  // user-sharded replica: users[b] is a row of this replica only for b in [seg[0], seg[1])
  for (int r = blockIdx.x * per + threadIdx.x / lanes; r < 3 * B; r += gridDim.x * per) {
    const int which = r / B, b = r - which * B;
    if (which == 0 && (b < s0 || b >= s1)) continue;
    const int64_t row = which == 0 ? users[b] : (int64_t)U + (which == 1 ? pos[b] : neg[b]);
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    bf16x8v o;
    if (src) {
      const float4 a = *reinterpret_cast<const float4*>(src + row * D + l * 8), c = *reinterpret_cast<const float4*>(src + row * D + l * 8 + 4);
      o = bf16x8v{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)c.x, (__bf16)c.y, (__bf16)c.z, (__bf16)c.w};
    } else {
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8v*>(dst + row * D + l * 8) = o;
  }
}
// The same rows of up to 8 tensors of any element size cleared: row_bytes[t] bytes per row (a multiple of 16)
struct BatchRowsZeroP { char* dst[8]; int row_bytes[8]; };
__global__ __launch_bounds__(256) void zero_batch_rows_kernel(BatchRowsZeroP P, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
                                                              const int32_t* __restrict__ neg, int B, int U, const int32_t* __restrict__ seg) {
  char* __restrict__ dst = P.dst[blockIdx.y];
  const int rb = P.row_bytes[blockIdx.y], lanes = rb / 16;  // 16-byte pieces per row
  const int s0 = seg ? seg[0] : 0, s1 = seg ? seg[1] : B;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)3 * B * lanes; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / lanes), l = (int)(e % lanes);
    const int which = r / B, b = r - which * B;
    if (which == 0 && (b < s0 || b >= s1)) continue;
    const int64_t row = which == 0 ? users[b] : (int64_t)U + (which == 1 ? pos[b] : neg[b]);
    *reinterpret_cast<uint4*>(dst + row * rb + (int64_t)l * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
}
extern "C" int dmm_zero_batch_rows(void* const* dst, const int32_t* row_bytes, int n_t, const int32_t* users, const int32_t* pos, const int32_t* neg, int B,
                                   int U, const int32_t* seg, void* stream) {
  DMM_CHECK_ARG(dst && row_bytes && n_t >= 1 && n_t <= 8 && users && pos && neg && B > 0 && U >= 0, "dmm_zero_batch_rows: bad arguments");
  BatchRowsZeroP P = {};
  int max_lanes = 0;
  for (int t = 0; t < n_t; ++t) {
    DMM_CHECK_ARG(dst[t] && ((uintptr_t)dst[t] & 15) == 0 && row_bytes[t] > 0 && row_bytes[t] % 16 == 0, "dmm_zero_batch_rows: tensor %d missing, misaligned or rows not whole 16-byte pieces", t);
    P.dst[t] = (char*)dst[t];
    P.row_bytes[t] = row_bytes[t];
    max_lanes = std::max(max_lanes, row_bytes[t] / 16);
  }
  const int64_t work = (int64_t)3 * B * max_lanes;
  hipLaunchKernelGGL(zero_batch_rows_kernel, dim3((unsigned)std::min<int64_t>(4096, (work + 255) / 256), n_t), dim3(256), 0, (hipStream_t)stream, P, users, pos, neg, B, U, seg);
  DMM_LAUNCH_CHECK("dmm_zero_batch_rows");
  return 0;
}
extern "C" int dmm_cast_batch_rows_bf16(const float* const* src, void* const* dst, int n_t, int D, const int32_t* users, const int32_t* pos, const int32_t* neg,
                                        int B, int U, const int32_t* seg, void* stream) {
  DMM_CHECK_ARG(dst && n_t >= 1 && n_t <= 4 && users && pos && neg && B > 0 && U >= 0, "dmm_cast_batch_rows_bf16: bad arguments");
  DMM_CHECK_ARG(D >= 8 && D <= 2048 && D % 8 == 0 && 256 % (D / 8) == 0, "dmm_cast_batch_rows_bf16: D=%d unsupported (a multiple of 8 with 256 %% (D / 8) == 0)", D);
  BatchRowsP P = {};
  for (int t = 0; t < n_t; ++t) {
    DMM_CHECK_ARG(dst[t] && ((uintptr_t)dst[t] & 15) == 0 && (!src || !src[t] || ((uintptr_t)src[t] & 15) == 0), "dmm_cast_batch_rows_bf16: tensor %d missing or misaligned", t);
    P.src[t] = src ? src[t] : nullptr;
    P.dst[t] = (__bf16*)dst[t];
  }
  const int per = 256 / (D / 8);
  hipLaunchKernelGGL(cast_batch_rows_kernel, dim3(std::min(2048, (3 * B + per - 1) / per), n_t), dim3(256), 0, (hipStream_t)stream, P, D, users, pos, neg, B, U, seg);
  DMM_LAUNCH_CHECK("dmm_cast_batch_rows_bf16");
  return 0;
}

extern "C" int dmm_segment_sum(const float* in, int n_seg, int64_t len, float scale, float* out, void* stream) {
  DMM_CHECK_ARG(in && out && n_seg > 0 && len >= 0, "dmm_segment_sum: bad arguments");
  hipLaunchKernelGGL(segment_sum_kernel, dim3(n_seg), dim3(256), 0, (hipStream_t)stream, in, len, scale, out);
  DMM_LAUNCH_CHECK("dmm_segment_sum");
  return 0;
}
extern "C" int64_t dmm_sumsq_scratch(int64_t n) { return grid_for(n / 4 + 1, 1024); }
extern "C" int dmm_sumsq(const float* x, int64_t n, float scale, float* scratch, float* out, void* stream) {
  DMM_CHECK_ARG(x && scratch && out && n > 0, "dmm_sumsq: bad arguments");
  DMM_CHECK_ARG(((uintptr_t)x & 15) == 0, "dmm_sumsq: x must be 16-byte aligned");
  const unsigned grid = grid_for(n / 4 + 1, 1024);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
  hipLaunchKernelGGL(segment_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, (int64_t)grid, scale, out);
  DMM_LAUNCH_CHECK("dmm_sumsq");
  return 0;
}

extern "C" int dmm_loss_reduce(const float* seg_in, int n_seg, int64_t len, float seg_scale, float* seg_out, const float* x, int64_t n, float x_scale,
                               float* scratch, float* x_out, double* acc, int acc_x, void* stream) {
  DMM_CHECK_ARG(seg_in && seg_out && n_seg > 0 && n_seg <= 65535 && len > 0 && x && scratch && x_out && n > 0 && ((uintptr_t)x & 15) == 0 && acc_x >= 0,
                "dmm_loss_reduce: bad arguments");
  const unsigned grid = grid_for(n / 4 + 1, 1024);  // scratch: dmm_sumsq_scratch(n) + 1 floats; the last one is the arrival counter (zero before the first call)
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(grid + n_seg), dim3(256), 0, (hipStream_t)stream, seg_in, n_seg, len, seg_scale, seg_out, x, n, x_scale, scratch,
                     reinterpret_cast<int*>(scratch + grid), x_out, acc, acc_x);
  DMM_LAUNCH_CHECK("dmm_loss_reduce");
  return 0;
}

extern "C" int dmm_zero_regions(void* const* ptrs, const int64_t* bytes, int n, void* stream) {
  DMM_CHECK_ARG(ptrs && bytes && n >= 1 && n <= 4, "dmm_zero_regions: 1..4 regions");
  ZeroRegions Z;
  memset(&Z, 0, sizeof(Z));
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    DMM_CHECK_ARG(ptrs[i] && bytes[i] >= 0 && bytes[i] % 16 == 0 && ((uintptr_t)ptrs[i] & 15) == 0, "dmm_zero_regions: region %d must be 16-byte aligned, whole 16-byte pieces", i);
    Z.p[i] = reinterpret_cast<float4*>(ptrs[i]);
    Z.n16[i] = bytes[i] / 16;
    most = std::max(most, Z.n16[i]);
  }
  Z.cnt = n;
  if (most == 0) return 0;
  hipLaunchKernelGGL(zero_regions_kernel, dim3(grid_for(most, 2048)), dim3(256), 0, (hipStream_t)stream, Z);
  DMM_LAUNCH_CHECK("dmm_zero_regions");
  return 0;
}

extern "C" int dmm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra, const int64_t* extra_n,
                             int n_extra, float l2_coef, float lr, float beta1, float beta2, float eps, int step, void* stream) {
  return dmm_adam_step_prefix(p, g, m, v, n, extra, extra_n, n_extra, l2_coef, lr, beta1, beta2, eps, step, stream, n);
}

extern "C" int dmm_adam_step_prefix(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra, const int64_t* extra_n,
                                    int n_extra, float l2_coef, float lr, float beta1, float beta2, float eps, int step, void* stream, int64_t l2_n) {
  return dmm_adam_step_x16(p, g, m, v, n, extra, extra_n, n_extra, l2_coef, lr, beta1, beta2, eps, step, stream, l2_n, nullptr, 0, 0);
}

extern "C" int dmm_adam_step_x16(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra, const int64_t* extra_n,
                                 int n_extra, float l2_coef, float lr, float beta1, float beta2, float eps, int step, void* stream, int64_t l2_n,
                                 void* p16, int64_t p16_n, int64_t p16_lo) {
  DMM_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && n_extra >= 0 && n_extra <= 4 && l2_n >= 0, "dmm_adam_step: bad arguments");
  DMM_CHECK_ARG(p16_n >= 0 && p16_lo >= 0 && p16_lo + p16_n <= n && (p16_n == 0 || p16), "dmm_adam_step_x16: p16 must cover the window [p16_lo, p16_lo + p16_n) of the n parameters");
  AdamExtra ex = {};
  ex.cnt = n_extra;
  for (int k = 0; k < n_extra; ++k) {
    ex.g[k] = extra[k];
    ex.n[k] = extra_n[k];
  }
  const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, ex, l2_coef,
                     (float)((double)lr / c1), beta1, beta2, (float)(1.0 / sqrt(c2)), eps, l2_n, (__bf16*)p16, p16_n, p16_lo);
  DMM_LAUNCH_CHECK("dmm_adam_step");
  return 0;
}

extern "C" int dmm_mul_sigmoid_fwd(const float* proj, const float* gate, float* out, int64_t n, void* stream) {
  DMM_CHECK_ARG(proj && gate && out && n > 0, "dmm_mul_sigmoid_fwd: bad arguments");
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, proj, gate, out, n);
  DMM_LAUNCH_CHECK("dmm_mul_sigmoid_fwd");
  return 0;
}
extern "C" int dmm_gate_bwd(const float* dout, const float* proj, const float* gate, float* dproj, float* dgate_pre, int64_t n, void* stream) {
  DMM_CHECK_ARG(dout && proj && gate && dproj && dgate_pre && n > 0, "dmm_gate_bwd: bad arguments");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, dout, proj, gate, dproj, dgate_pre, n);
  DMM_LAUNCH_CHECK("dmm_gate_bwd");
  return 0;
}
extern "C" int dmm_tanh_bwd(const float* dh, const float* h, float* dpre, int64_t n, void* stream) {
  DMM_CHECK_ARG(dh && h && dpre && n > 0, "dmm_tanh_bwd: bad arguments");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, dh, h, dpre, n);
  DMM_LAUNCH_CHECK("dmm_tanh_bwd");
  return 0;
}
extern "C" int dmm_bias_tanh_rows(float* x, const float* bias, int rows, int cols, void* out16, void* stream) {
  DMM_CHECK_ARG(x && bias && rows > 0 && cols > 0 && cols % 4 == 0 && (((uintptr_t)x | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)out16 & 7) == 0,
                "dmm_bias_tanh_rows: bad arguments (contiguous [rows, cols], cols a multiple of 4, 16-byte aligned)");
  const int64_t n4 = (int64_t)rows * cols / 4;
  hipLaunchKernelGGL(bias_tanh_rows_kernel, dim3(grid_for(n4, 1024)), dim3(256), 0, (hipStream_t)stream, x, bias, cols, n4, (__bf16*)out16);
  DMM_LAUNCH_CHECK("dmm_bias_tanh_rows");
  return 0;
}
extern "C" int dmm_add_inplace(float* y, const float* x, float alpha, int64_t n, void* stream) {
  DMM_CHECK_ARG(y && x && n > 0, "dmm_add_inplace: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, n);
  DMM_LAUNCH_CHECK("dmm_add_inplace");
  return 0;
}
static int colsum_launch(const char* who, const float* x, int64_t ldx, int rows, int cols, float* out, bool zero_first, void* stream) {
  DMM_CHECK_ARG(x && out && rows > 0 && cols > 0, "%s: bad arguments", who);
  const int chunk = rows > 64 ? 64 : rows;
  if (zero_first) (void)hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), cdiv(rows, chunk)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, chunk, out);
  DMM_LAUNCH_CHECK(who);
  return 0;
}
extern "C" int dmm_colsum(const float* x, int64_t ldx, int rows, int cols, float* out, void* stream) {
  return colsum_launch("dmm_colsum", x, ldx, rows, cols, out, true, stream);
}
extern "C" int dmm_colsum_add(const float* x, int64_t ldx, int rows, int cols, float* out, void* stream) {
  return colsum_launch("dmm_colsum_add", x, ldx, rows, cols, out, false, stream);
}

// One uniform negative per interaction, exactly uniform over the items the user has NOT interacted with (TrainData.negSampling, DataHandler.py:159-169:
// the distribution the reference's rejection loop samples), in one pass and without rejection: with the user's k positives p_0 < ... < p_{k-1}, draw
// r uniformly from [0, I - k) and return the r-th non-positive item r + #{j : p_j - j <= r} (p_j - j, the number of non-positives below p_j, is
// non-decreasing: a binary search).  Philox4x32-10 keyed by (seed, offset + slot): no generator state, no host round trip.
__global__ __launch_bounds__(256) void neg_sample_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const int32_t* __restrict__ users,
                                                          int n, int I, unsigned long long seed, unsigned long long offset, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int u = users[i];
  const int64_t b = ptr[u];
  const int k = (int)(ptr[u + 1] - b);
  const int m = I - k;
  if (m <= 0) { out[i] = 0; return; }  // (a user who interacted with every item has no negative: the reference would loop forever)
  const uint4 rnd = dmm::philox4(seed, offset + (unsigned long long)i, 0x6e65u);
  const unsigned long long wide = ((unsigned long long)rnd.x << 32) | rnd.y;
  const int r = (int)(((unsigned __int128)wide * (unsigned long long)m) >> 64);  // uniform in [0, m) (bias < 2^-32)
  int lo = 0, hi = k;  // first j with col[b + j] - j > r
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[b + mid] - mid <= r) lo = mid + 1; else hi = mid;
  }
  out[i] = r + lo;
}

extern "C" int dmm_neg_sample(const int64_t* ptr, const int32_t* col_sorted, const int32_t* users, int n, int items, uint64_t seed, uint64_t offset, int32_t* out,
                              void* stream) {
  DMM_CHECK_ARG(ptr && col_sorted && users && out && n > 0 && items > 0, "dmm_neg_sample: bad arguments");
  hipLaunchKernelGGL(neg_sample_kernel, dim3(dmm::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ptr, col_sorted, users, n, items, (unsigned long long)seed,
                     (unsigned long long)offset, out);
  DMM_LAUNCH_CHECK("dmm_neg_sample");
  return 0;
}

extern "C" int dmm_fill_zero(void* p, int64_t bytes, void* stream) {
  DMM_CHECK_ARG(p && bytes >= 0, "dmm_fill_zero: bad arguments");
  hipError_t err = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  if (err != hipSuccess) {
    dmm::set_error("dmm_fill_zero: %s", hipGetErrorString(err));
    return (int)err;
  }
  return 0;
}
