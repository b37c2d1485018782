// Shared between gemm_bf16.hip (register-staged tiles) and gemm_bf16_8ph.hip (256 x 256 LDS-DMA tile).
#pragma once
#include "common.h"

namespace dmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmBf16P {
  const __bf16* A; long long lda;
  const __bf16* B; long long ldb;
  float* C; long long ldc;
  __bf16* Cb; long long ldcb;
  const float* bias;
  const float* bias2;  // non-null: Cb = tanh(C + bias2[n]) instead of a copy of C (needs C and Cb)
  const float* R; long long ldr;
  const __bf16* Rb; long long ldrb;  // residual given in bf16 (alternative to R)
  int M, N, K;
  float alpha, e1, e2;
  int act;
  int mt, nt;  // tile counts
  int part;    // XCD placement: 0 = grid order; register-staged tiles: 1 = each XCD owns a range of m tiles, 2 = a range of n tiles;
               // 256x256 tile: 1 = XCD-contiguous row-major ids, 2 = XCD-contiguous supertiles (default)
};

constexpr int GB_BK = 64;
constexpr int GB_MAX_BATCH = 4;

struct GemmBf16Batch {
  GemmBf16P p[GB_MAX_BATCH];  // same shape / scalars, different pointers (one per modality); blockIdx.y picks one
  int n;
};

// tanh for outputs that are rounded to bf16 anyway: 1 - 2 / (2^(2x log2 e) + 1), two transcendental instructions
__device__ __forceinline__ float tanh_fast_bf16(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// 256 x 256 tile kernel (gemm_bf16_8ph.hip); requires K % 64 == 0.  p[i].mt / nt must be the 256-tile counts.
void launch_gemm_bf16_8ph(const GemmBf16Batch& b, hipStream_t s);

}  // namespace dmm
