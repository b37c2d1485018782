// Contrastive (InfoNCE) and ranking (BPR) losses, forward and backward, plus the gather / scatter-add of the
// minibatch rows.  Replaces Utils/Utils.py:57-98 and the index_select / index_add autograd pairs of
// Main.py:305-307 and Utils.py:67-68.
//
// InfoNCE never materialises the B x B logit matrix: one wave owns one anchor row, keeps its own normalised
// row in LDS, streams the other view's rows out of L2 and folds them into an online log-sum-exp (forward) or
// into a probability row held in LDS (backward).  [v1: fp32 VALU dot products; the MFMA tiling is the next step.]
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
template <int D>
__global__ __launch_bounds__(256) void gather_normalize_kernel(SlotTable S, int B, float* __restrict__ zhat, float* __restrict__ norm) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int s = blockIdx.y;
  const int l = threadIdx.x % LANES;
  const float* __restrict__ src = S.src[s];
  const int32_t* __restrict__ idx = S.idx[s];
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < B; b += gridDim.x * G) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)idx[b] * D + 4 * l);
    const float nrm = sqrtf(group_sum<LANES>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    *reinterpret_cast<float4*>(zhat + ((size_t)s * B + b) * D + 4 * l) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    if (l == 0) norm[(size_t)s * B + b] = nrm;
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
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < B; b += gridDim.x * G) {
    const size_t off = ((size_t)s * B + b) * D + 4 * l;
    const float4 g = *reinterpret_cast<const float4*>(dzhat + off);
    const float4 y = *reinterpret_cast<const float4*>(zhat + off);
    const float nrm = norm[(size_t)s * B + b];
    float4 o;
    if (nrm < 1e-12f) {
      o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    } else {
      const float dot = group_sum<LANES>(g.x * y.x + g.y * y.y + g.z * y.z + g.w * y.w);
      const float inv = 1.0f / nrm;
      o = make_float4((g.x - y.x * dot) * inv, (g.y - y.y * dot) * inv, (g.z - y.z * dot) * inv, (g.w - y.w * dot) * inv);
    }
    float* dst = grad + (size_t)idx[b] * D + 4 * l;
    atomicAdd(dst + 0, o.x);
    atomicAdd(dst + 1, o.y);
    atomicAdd(dst + 2, o.z);
    atomicAdd(dst + 3, o.w);
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
// BPR forward + backward
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void bpr_kernel(const float* __restrict__ fin, const int32_t* __restrict__ users,
                                                   const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int B, int U,
                                                   float* __restrict__ row_loss, float* __restrict__ dfin) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int l = threadIdx.x % LANES;
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * G + threadIdx.x / LANES; b < B; b += gridDim.x * G) {
    const size_t ou = (size_t)users[b] * D + 4 * l, op = (size_t)(U + pos[b]) * D + 4 * l, on = (size_t)(U + neg[b]) * D + 4 * l;
    const float4 u = *reinterpret_cast<const float4*>(fin + ou);
    const float4 p = *reinterpret_cast<const float4*>(fin + op);
    const float4 n = *reinterpret_cast<const float4*>(fin + on);
    const float sp = group_sum<LANES>(u.x * p.x + u.y * p.y + u.z * p.z + u.w * p.w);
    const float sn = group_sum<LANES>(u.x * n.x + u.y * n.y + u.z * n.z + u.w * n.w);
    const float sg = 1.0f / (1.0f + expf(-(sp - sn)));
    if (l == 0) row_loss[b] = -logf(1e-5f + sg) * invB;
    if (dfin) {
      const float g = -(sg * (1.0f - sg)) / (1e-5f + sg) * invB;  // d row_loss / d (sp - sn)
      atomicAdd(dfin + ou + 0, g * (p.x - n.x)); atomicAdd(dfin + ou + 1, g * (p.y - n.y));
      atomicAdd(dfin + ou + 2, g * (p.z - n.z)); atomicAdd(dfin + ou + 3, g * (p.w - n.w));
      atomicAdd(dfin + op + 0, g * u.x); atomicAdd(dfin + op + 1, g * u.y);
      atomicAdd(dfin + op + 2, g * u.z); atomicAdd(dfin + op + 3, g * u.w);
      atomicAdd(dfin + on + 0, -g * u.x); atomicAdd(dfin + on + 1, -g * u.y);
      atomicAdd(dfin + on + 2, -g * u.z); atomicAdd(dfin + on + 3, -g * u.w);
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

extern "C" int dmm_gather_normalize(const dmm_gather_desc* slots, int n_slots, int B, int D, float* zhat, float* norm, void* stream) {
  SlotTable S;
  if (int e = fill_slots(slots, n_slots, S)) return e;
  DMM_CHECK_ARG(zhat && norm && B > 0 && d_ok(D), "dmm_gather_normalize: bad arguments (B=%d D=%d)", B, D);
  const int G = 1024 / D;
  dim3 grid(cdiv(B, G) > 1024 ? 1024 : cdiv(B, G), n_slots);
  DISPATCH_D(D, hipLaunchKernelGGL(gather_normalize_kernel<DD>, grid, dim3(256), 0, (hipStream_t)stream, S, B, zhat, norm));
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

extern "C" int dmm_infonce_fwd(const float* zhat, const dmm_infonce_term* terms, int n_terms, int B, int D, float* lse, float* row_loss,
                               void* stream) {
  TermTable T;
  if (int e = fill_terms(terms, n_terms, T)) return e;
  DMM_CHECK_ARG(zhat && lse && row_loss && B > 0 && D % 4 == 0 && D > 0, "dmm_infonce_fwd: bad arguments");
  const int w = infonce_waves(B, D, false);
  DMM_CHECK_ARG(w > 0, "dmm_infonce_fwd: D=%d too large", D);
  hipLaunchKernelGGL(infonce_fwd_kernel, dim3(cdiv(B, w), n_terms), dim3(64 * w), (size_t)w * D * sizeof(float), (hipStream_t)stream,
                     zhat, T, B, D, lse, row_loss);
  DMM_LAUNCH_CHECK("dmm_infonce_fwd");
  return 0;
}

extern "C" int dmm_infonce_bwd(const float* zhat, const float* lse, const dmm_infonce_term* terms, int n_terms, int B, int D, float* dzhat,
                               void* stream) {
  TermTable T;
  if (int e = fill_terms(terms, n_terms, T)) return e;
  DMM_CHECK_ARG(zhat && lse && dzhat && B > 0 && D % 4 == 0 && D > 0, "dmm_infonce_bwd: bad arguments");
  const int w = infonce_waves(B, D, true);
  DMM_CHECK_ARG(w > 0, "dmm_infonce_bwd: batch %d too large for the v1 kernel (LDS row buffer)", B);
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3(cdiv(B, w), n_terms, 2), dim3(64 * w), (size_t)w * ((D + B + 3) & ~3) * sizeof(float),
                     (hipStream_t)stream, zhat, lse, T, B, D, dzhat);
  DMM_LAUNCH_CHECK("dmm_infonce_bwd");
  return 0;
}

extern "C" int dmm_bpr_fwd_bwd(const float* fin, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int D, int U,
                               float* row_loss, float* dfin, void* stream) {
  DMM_CHECK_ARG(fin && users && pos && neg && row_loss && B > 0 && d_ok(D), "dmm_bpr_fwd_bwd: bad arguments");
  const int G = 1024 / D;
  const unsigned grid = cdiv(B, G) > 1024 ? 1024 : cdiv(B, G);
  DISPATCH_D(D, hipLaunchKernelGGL(bpr_kernel<DD>, dim3(grid), dim3(256), 0, (hipStream_t)stream, fin, users, pos, neg, B, U, row_loss, dfin));
  DMM_LAUNCH_CHECK("dmm_bpr_fwd_bwd");
  return 0;
}
