// Shared device/host helpers for libdiffmm_hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/diffmm_hip.h"

namespace dmm {

void set_error(const char* fmt, ...);

#define DMM_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      dmm::set_error(__VA_ARGS__);               \
      return DMM_ERR_INVALID_ARG;                \
    }                                            \
  } while (0)

// Launch check: kernels are asynchronous, so this only catches launch-configuration errors.
#define DMM_LAUNCH_CHECK(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      dmm::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));          \
      return (int)e__;                                                                \
    }                                                                                 \
  } while (0)

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (Salmon et al. 2011).  key = seed, counter = (index, stream).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ uint4 philox4(unsigned long long seed, unsigned long long idx, uint32_t stream) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  return make_uint4(c[0], c[1], c[2], c[3]);
}

// uniform in [0,1) with 24 bits, like torch.rand for float32
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// two standard normals from two uniforms (Box-Muller)
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
  float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
  float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.283185307179586f * u2, &s, &c);
  return make_float2(r * c, r * s);
}

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

// reduce over `width` consecutive lanes (width = power of two <= 64, aligned groups)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[w] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}
__device__ __forceinline__ double block_sum_256d(double v, double* sm4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[w] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// bijective XCD-aware remap: blocks that share an XCD (b % 8 under round-robin dispatch) get a contiguous
// range of logical ids, so each XCD's private L2 serves one contiguous slice of the work list.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

}  // namespace dmm
