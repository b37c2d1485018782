// CSR SpMM  Y = alpha * A X (+ beta C1 + gamma C2)  with fused epilogues, batched over up to 4 problems.
//
// Replaces the reference's torch.sparse.mm call sites (Model.py:90,93,105,111,114,123,130; Main.py:319)
// and their autograd backward (A symmetric => dX = A dY, the same kernel).
//
// Work decomposition (MI355X: 256 CUs x 4 SIMDs, wave = 64):
//   * one "group" of LANES = D/4 consecutive lanes owns one TASK = (row, [nz_begin, nz_end)); every lane
//     keeps a float4 of the output row, so one group reads a full X row as one coalesced D*4-byte segment;
//   * rows longer than the planner's task length are split into k <= G consecutive tasks of the SAME
//     workgroup (G = 1024 / D groups per 256-thread workgroup); the partial rows are combined through LDS in
//     task order, so results are bitwise reproducible -- no atomics.  This is what balances the TikTok
//     degree law (1 % of the rows hold 70 % of the non-zeros) across waves;
//   * the host planner (diffmm_amd/graph.py: plan_tasks) emits the task list; padding tasks have row = -1.
//   * workgroups keep the dispatcher's round-robin order over the 8 XCDs; an XCD-contiguous remap of the task list exists
//     (DMM_SPMM_REMAP=1) but measured neutral at TikTok size and 20-30 % slower at 1M users (unequal work per XCD);
//   * rows beyond one workgroup's worth of tasks are cut into segments (several workgroups) that a tiny second launch sums.
//
// Epilogues (all optional, selected per problem):
//   y      = alpha*acc + beta*c1[row] + gamma*c2[row]
//   y_pert = y + sign(y) * normalize_row(noise[row]) * noise_degree      (Main.py:320-321; noise read from a
//            buffer for parity tests, or drawn in-kernel with Philox when noise == nullptr)
//   y_mean = (y_pert + m1[row] + m2[row]) / 3                             (Main.py:325, mean of the 3 CL layers)
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace dmm {

struct SpmmBatch {
  dmm_spmm_problem p[DMM_SPMM_MAX_BATCH];
  int block_begin[DMM_SPMM_MAX_BATCH + 1];
  int n;
  int remap;
};

// A lane owns VEC consecutive columns of a row = one 16-byte piece of the X row it gathers: VEC = 4 for fp32 operands, 8 for bf16
// operands (dmm_spmm_csr_bf16: half the gather bytes, twice the tasks per wave).  Everything after the gather is fp32.
template <int VEC>
struct Acc {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Acc<VEC> load_f32(const float* p) {
  Acc<VEC> a;
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const float4 t = reinterpret_cast<const float4*>(p)[q];
    a.v[4 * q] = t.x; a.v[4 * q + 1] = t.y; a.v[4 * q + 2] = t.z; a.v[4 * q + 3] = t.w;
  }
  return a;
}
template <int VEC>
__device__ __forceinline__ void store_f32(float* p, const Acc<VEC>& a) {
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) reinterpret_cast<float4*>(p)[q] = make_float4(a.v[4 * q], a.v[4 * q + 1], a.v[4 * q + 2], a.v[4 * q + 3]);
}
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void store_bf16(void* base, size_t elem, const Acc<VEC>& a) {
  __bf16* p = reinterpret_cast<__bf16*>(base) + elem;
  if (VEC == 8) {
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)a.v[j % VEC];
    *reinterpret_cast<bf16x8_t*>(p) = o;
  } else {
    bf16x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)a.v[j];
    *reinterpret_cast<bf16x4_t*>(p) = o;
  }
}

template <int D, int VEC>
__device__ __forceinline__ void spmm_epilogue(const dmm_spmm_problem& P, int row, int l, const Acc<VEC>& acc) {
  constexpr int LANES = D / VEC;
  const size_t off = (size_t)row * D + VEC * l;
  Acc<VEC> y;
#pragma unroll
  for (int j = 0; j < VEC; ++j) y.v[j] = P.alpha * acc.v[j];
  if (P.c1) {
    const Acc<VEC> c = load_f32<VEC>(P.c1 + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) y.v[j] = fmaf(P.beta, c.v[j], y.v[j]);
  }
  if (P.c2) {
    const Acc<VEC> c = load_f32<VEC>(P.c2 + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) y.v[j] = fmaf(P.gamma, c.v[j], y.v[j]);
  }
  if (P.y) store_f32<VEC>(P.y + off, y);
  if (P.y16) store_bf16<VEC>(P.y16, off, y);
  if (P.y_pert || P.y_pert16) {
    Acc<VEC> n;
    if (P.noise) {
      n = load_f32<VEC>(P.noise + off);
    } else {
      // the counter uses the row's GLOBAL id, so a user-sharded replica draws exactly what the unsharded launch draws for that row;
      // element c of the row <-> counter grow * D/4 + c/4, component c % 4, whatever the lane width
      const unsigned long long grow = (unsigned long long)(row + (row < P.rng_row_split ? P.rng_row_add_lo : P.rng_row_add_hi));
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const uint4 r = philox4(P.seed, P.offset + grow * (D / 4) + (VEC / 4) * l + q, 0x5eedu);
        n.v[4 * q] = u01(r.x); n.v[4 * q + 1] = u01(r.y); n.v[4 * q + 2] = u01(r.z); n.v[4 * q + 3] = u01(r.w);
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) ss = fmaf(n.v[j], n.v[j], ss);
    ss = group_sum<LANES>(ss);
    const float sc = P.noise_degree / fmaxf(sqrtf(ss), 1e-12f);
    Acc<VEC> yp;
#pragma unroll
    for (int j = 0; j < VEC; ++j) yp.v[j] = y.v[j] + sgn(y.v[j]) * (n.v[j] * sc);
    if (P.y_pert) store_f32<VEC>(P.y_pert + off, yp);
    if (P.y_pert16) store_bf16<VEC>(P.y_pert16, off, yp);
    if (P.y_mean) {
      const Acc<VEC> a = load_f32<VEC>(P.m1 + off), b = load_f32<VEC>(P.m2 + off);
      const float t = 1.0f / 3.0f;
      Acc<VEC> m;
#pragma unroll
      for (int j = 0; j < VEC; ++j) m.v[j] = (yp.v[j] + a.v[j] + b.v[j]) * t;
      store_f32<VEC>(P.y_mean + off, m);
    }
  }
}

// User-sharded layout: rows >= raw_from (the replicated item rows) are only PARTIAL sums on a replica -- the kernel parks the
// raw accumulator in `raw` ([n_rows - raw_from, D]); after the replicas' partials are summed (one all-reduce per launch),
// spmm_epilogue_rows_kernel applies the epilogue to those rows.
template <int D, int VEC>
__device__ __forceinline__ void spmm_finish(const dmm_spmm_problem& P, int row, int l, const Acc<VEC>& acc) {
  if (P.raw && row >= P.raw_from) {
    store_f32<VEC>(P.raw + (size_t)(row - P.raw_from) * D + VEC * l, acc);
    return;
  }
  spmm_epilogue<D, VEC>(P, row, l, acc);
}

template <int D>
__global__ __launch_bounds__(256) void spmm_epilogue_rows_kernel(const SpmmBatch batch) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const dmm_spmm_problem& P = batch.p[blockIdx.y];
  if (!P.raw) return;
  // a plain problem (y = A x, nothing else) whose raw rows were parked in y's own rows and summed there: nothing is left to do
  if (P.y && P.raw == P.y + (size_t)P.raw_from * D && P.alpha == 1.0f && !P.c1 && !P.c2 && !P.y16 && !P.y_pert && !P.y_pert16 && !P.y_mean) return;
  const int l = threadIdx.x % LANES;
  for (int row = P.raw_from + blockIdx.x * G + threadIdx.x / LANES; row < P.n_rows; row += gridDim.x * G) {
    const Acc<4> acc = load_f32<4>(P.raw + (size_t)(row - P.raw_from) * D + 4 * l);
    spmm_epilogue<D, 4>(P, row, l, acc);
  }
}

template <int D>
__device__ __forceinline__ void spmm_fix_row(const dmm_spmm_problem& P, int fi, float4* part);

// VEC 4 = fp32 operands, VEC 8 = bf16 operands; DEPTH = X-row gathers in flight per lane; MINW = waves per SIMD the register
// allocation must leave room for (the kernel is bound by the number of tasks in flight: tools/spmm_scale.py)
template <int D, int VEC, int DEPTH, int MINW>
__global__ __launch_bounds__(256, MINW) void spmm_kernel(const SpmmBatch batch) {
  constexpr int LANES = D / VEC;    // lanes per task group
  constexpr int G = 256 / LANES;    // task groups per workgroup
  __shared__ float part[256 * VEC];

  // which problem does this workgroup belong to (n <= 4: a short uniform scan)
  int pi = 0;
#pragma unroll
  for (int i = 1; i < DMM_SPMM_MAX_BATCH; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.block_begin[i]) pi = i;
  const dmm_spmm_problem& P = batch.p[pi];
  const int nblk = batch.block_begin[pi + 1] - batch.block_begin[pi];
  const int head = P.n_head_tasks / G;  // workgroups of split heavy rows: spread over the XCDs by the dispatcher itself
  int lb = (int)blockIdx.x - batch.block_begin[pi];
  if (batch.remap && lb >= head) lb = head + xcd_remap(lb - head, nblk - head);

  const int g = threadIdx.x / LANES;
  const int l = threadIdx.x % LANES;
  const int4 task = reinterpret_cast<const int4*>(P.tasks)[lb * G + g];
  const int row = task.x, beg = task.y, end = task.z;
  const int run_len = task.w & 0xffff, run_pos = task.w >> 16;

  Acc<VEC> acc;
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
  if (row >= 0) {
    const int* __restrict__ col = P.col;
    const float* __restrict__ val = P.val;
    const char* __restrict__ xlo = reinterpret_cast<const char*>(P.x_lo);
    const char* __restrict__ xhi = reinterpret_cast<const char*>(P.x_hi);
    const int split = P.split;
    constexpr size_t ROWB = (size_t)D * (16 / VEC);  // bytes per X row
    // The group's lanes fetch LANES (column, value) pairs with one coalesced load each, then every pair is broadcast
    // inside the group (ds_bpermute) and 8 independent X-row gathers are kept in flight per lane: a 32-non-zero task costs
    // ~5 dependent memory round trips instead of 16.
    for (int base = beg; base < end; base += LANES) {
      const int cnt = min(LANES, end - base);
      int myc = 0;
      float myv = 0.f;
      if (l < cnt) {
        myc = col[base + l];
        myv = val[base + l];
      }
      for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
        uint4 x[DEPTH];
        float v[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
          const int c = __shfl(myc, j0 + j, LANES);
          v[j] = __shfl(myv, j0 + j, LANES);
          x[j] = make_uint4(0u, 0u, 0u, 0u);
          if (j0 + j < cnt) {
            const char* src = c < split ? xlo + (size_t)c * ROWB : xhi + (size_t)(c - split) * ROWB;
            x[j] = reinterpret_cast<const uint4*>(src)[l];
          } else {
            v[j] = 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
          if (VEC == 4) {
            acc.v[0] = fmaf(v[j], __uint_as_float(x[j].x), acc.v[0]);
            acc.v[1] = fmaf(v[j], __uint_as_float(x[j].y), acc.v[1]);
            acc.v[2 % VEC] = fmaf(v[j], __uint_as_float(x[j].z), acc.v[2 % VEC]);
            acc.v[3 % VEC] = fmaf(v[j], __uint_as_float(x[j].w), acc.v[3 % VEC]);
          } else {  // 8 bf16 per 16 bytes: element 2k = low half of word k (bf16 -> fp32 is a 16-bit shift)
            const unsigned w[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              acc.v[(2 * k) % VEC] = fmaf(v[j], __uint_as_float(w[k] << 16), acc.v[(2 * k) % VEC]);
              acc.v[(2 * k + 1) % VEC] = fmaf(v[j], __uint_as_float(w[k] & 0xffff0000u), acc.v[(2 * k + 1) % VEC]);
            }
          }
        }
      }
    }
  }
  // combine the tasks of a split row in task order (deterministic)
#pragma unroll
  for (int j = 0; j < VEC; ++j) part[j * 256 + threadIdx.x] = acc.v[j];
  __syncthreads();
  const bool run_head = row >= 0 && run_pos == 0;
  if (run_head) {
    for (int t = 1; t < run_len; ++t) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc.v[j] += part[j * 256 + (g + t) * LANES + l];
    }
  }
  if (lb >= P.n_partial_blocks) {  // (uniform over the workgroup)
    if (run_head) spmm_finish<D, VEC>(P, row, l, acc);
    return;
  }
  // ---- segment of a giant row: park the raw partial row; the segments are summed in order by spmm_fixup_kernel (a second launch) or, with
  // fix_count, by whichever segment workgroup of the row arrives last (no second launch; the same order and grouping: bit-identical)
  if (run_head) store_f32<VEC>(P.partial + (size_t)lb * D + VEC * l, acc);
  if (!P.fix_count) return;
  __shared__ int s_fix;  // index of the giant row when this workgroup's arrival was the last one, else -1
  // publish (MI355X_MICROARCH.md, inter-workgroup visibility, R1): every storing wave drains its stores, workgroup barrier, ONE lane: agent release
  // (L2 write-back) -> wait -> agent-scope add; the last arriver: agent acquire -> wait -> (barrier) -> plain loads
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (also: every read of `part` above is done before spmm_fix_row reuses it)
  if (threadIdx.x == 0) {
    const int4* fix = reinterpret_cast<const int4*>(P.fix);
    int lo = 0, hi = P.n_fix - 1;  // the giant row whose slots [first_slot, first_slot + n_slots) hold lb (rows are listed by first_slot)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (fix[mid].y <= lb) lo = mid; else hi = mid - 1;
    }
    const int n_slots = fix[lo].z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait: the add must not overtake the write-back)
    const int old = __hip_atomic_fetch_add(P.fix_count + lo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int mine = -1;
    if (old == n_slots - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(P.fix_count + lo, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
      mine = lo;
    }
    s_fix = mine;
  }
  __syncthreads();
  const int fi = s_fix;
  if (fi < 0) return;
  spmm_fix_row<D>(P, fi, reinterpret_cast<float4*>(part));
}

// Rows longer than one workgroup's worth of tasks are cut into segments (one workgroup each, planned first in the task list).
// spmm_fix_row adds a row's segment partials and applies the epilogue: one WORKGROUP per giant row -- group g sums the segments
// g, g + G, g + 2G, ... and the G partial rows are then added in group order through LDS (a fixed order: deterministic).  It is run by
// the last-arriving segment workgroup of the main launch (dmm_spmm_problem.fix_count) or by spmm_fixup_kernel, a second launch.
// part = 256 float4 of LDS; all 256 threads of the workgroup must call it.
template <int D>
__device__ __forceinline__ void spmm_fix_row(const dmm_spmm_problem& P, int fi, float4* part) {
  constexpr int LANES = D / 4, G = 256 / LANES;
  const int g = threadIdx.x / LANES, l = threadIdx.x % LANES;
  const int4 f = reinterpret_cast<const int4*>(P.fix)[fi];  // {row, first_slot, n_slots, -}
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sidx = g; sidx < f.z; sidx += G) {
    const float4 o = *reinterpret_cast<const float4*>(P.partial + (size_t)(f.y + sidx) * D + 4 * l);
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
  }
  part[threadIdx.x] = a;
  __syncthreads();
  if (g != 0) return;
  for (int t = 1; t < G; ++t) {
    const float4 o = part[t * LANES + l];
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
  }
  Acc<4> acc;
  acc.v[0] = a.x; acc.v[1] = a.y; acc.v[2] = a.z; acc.v[3] = a.w;
  spmm_finish<D, 4>(P, f.x, l, acc);
}

template <int D>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const SpmmBatch batch) {
  __shared__ float4 part[256];
  const dmm_spmm_problem& P = batch.p[blockIdx.y];
  if ((int)blockIdx.x >= P.n_fix || P.fix_count) return;  // (fix_count: the main launch finished this problem's giant rows itself)
  spmm_fix_row<D>(P, blockIdx.x, part);
}

// ---- EXPERIMENT (round 4, VERDICT r3 item 9: "the LDS-staged ... variant north_star names, once, in-kernel, at 1M x 500k"): hot X rows in LDS --------
// A column-BLOCKED pass (stage a block of X rows, sweep the rows that reference it) multiplies the Y traffic by the number of blocks at this density
// (1e-5: priced in DESIGN 4.1, ~10 GB against 2.8 GB) -- what CAN be staged is the HEAD of the reference distribution: the n_hot most-referenced X rows
// (Zipf items, heavy users) are copied into each persistent workgroup's LDS once; `col` carries them as -1 - slot, every other gather goes to global
// memory as before.  1024-thread workgroups (four 256-thread quarters, each the original kernel's workgroup), one per CU (the hot table is 112 KiB);
// bf16 operands, d = 64, plain problems (y = A x) only.  Measured at 1M x 500k (profiles/r4_spmm_hot_lds.txt): see DESIGN 4.1.
constexpr int HOT_MAX = 896;
__global__ __launch_bounds__(1024) void spmm_hot_lds_kernel(const dmm_spmm_problem P, const int32_t* __restrict__ hot_ids, int n_hot,
                                                            const int32_t* __restrict__ col_hot, int nblocks) {
  constexpr int D = 64, VEC = 8, LANES = D / VEC, G = 256 / LANES;  // 8 lanes per task, 32 tasks per quarter
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  uint4* const hot = reinterpret_cast<uint4*>(hsm);                                   // [n_hot][8] 16-byte pieces
  float* const part_all = reinterpret_cast<float*>(hsm + (size_t)HOT_MAX * D * 2);     // [4][256 * VEC]
  const int quarter = threadIdx.x >> 8, tq = threadIdx.x & 255;
  float* const part = part_all + quarter * 256 * VEC;
  const int g = tq / LANES, l = tq % LANES;
  const char* __restrict__ xlo = reinterpret_cast<const char*>(P.x_lo);
  constexpr size_t ROWB = (size_t)D * 2;
  for (int s = threadIdx.x / LANES; s < n_hot; s += 1024 / LANES) hot[s * LANES + l] = reinterpret_cast<const uint4*>(xlo + (size_t)hot_ids[s] * ROWB)[l];
  __syncthreads();
  const float* __restrict__ val = P.val;
  for (int it = blockIdx.x; it * 4 < nblocks; it += gridDim.x) {
    const int lb = it * 4 + quarter;
    const bool live_q = lb < nblocks;
    int4 task = make_int4(-1, 0, 0, 1);
    if (live_q) task = reinterpret_cast<const int4*>(P.tasks)[lb * G + g];
    const int row = task.x, beg = task.y, end = task.z;
    const int run_len = task.w & 0xffff, run_pos = task.w >> 16;
    Acc<VEC> acc;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
    if (row >= 0) {
      for (int base = beg; base < end; base += LANES) {
        const int cnt = min(LANES, end - base);
        int myc = 0;
        float myv = 0.f;
        if (l < cnt) { myc = col_hot[base + l]; myv = val[base + l]; }
        uint4 x[8];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = __shfl(myc, j, LANES);
          v[j] = __shfl(myv, j, LANES);
          x[j] = make_uint4(0u, 0u, 0u, 0u);
          if (j < cnt) x[j] = c < 0 ? hot[(-1 - c) * LANES + l] : reinterpret_cast<const uint4*>(xlo + (size_t)c * ROWB)[l];
          else v[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned w[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc.v[2 * k] = fmaf(v[j], __uint_as_float(w[k] << 16), acc.v[2 * k]);
            acc.v[2 * k + 1] = fmaf(v[j], __uint_as_float(w[k] & 0xffff0000u), acc.v[2 * k + 1]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[j * 256 + tq] = acc.v[j];
    __syncthreads();
    if (row >= 0 && run_pos == 0) {
      for (int t = 1; t < run_len; ++t) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc.v[j] += part[j * 256 + (g + t) * LANES + l];
      }
      if (lb < P.n_partial_blocks) store_f32<VEC>(P.partial + (size_t)lb * D + VEC * l, acc);
      else store_f32<VEC>(P.y + (size_t)row * D + VEC * l, acc);
    }
    __syncthreads();
  }
}

extern "C" int dmm_spmm_csr_bf16_hotlds(const dmm_spmm_problem* prob, const int32_t* hot_ids, int n_hot, const int32_t* col_hot, int grid, void* stream) {
  DMM_CHECK_ARG(prob && hot_ids && col_hot && n_hot >= 0 && n_hot <= dmm::HOT_MAX && grid > 0, "dmm_spmm_csr_bf16_hotlds: bad arguments (n_hot <= %d)", dmm::HOT_MAX);
  const dmm_spmm_problem& p = *prob;
  DMM_CHECK_ARG(p.tasks && p.val && p.x_lo && p.y && !p.x_hi && !p.c1 && !p.c2 && !p.y_pert && !p.raw && p.alpha == 1.0f && p.n_tasks % 32 == 0,
                "dmm_spmm_csr_bf16_hotlds: one plain problem y = A x (bf16 x, fp32 y), d = 64");
  DMM_CHECK_ARG(p.n_partial_blocks == 0 || (p.partial && p.fix && p.n_fix > 0), "dmm_spmm_csr_bf16_hotlds: giant rows need partial + fix buffers");
  const size_t lds = (size_t)dmm::HOT_MAX * 64 * 2 + 4 * 256 * 8 * sizeof(float);
  static bool once = [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dmm::spmm_hot_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); return true; }();
  (void)once;
  hipLaunchKernelGGL(dmm::spmm_hot_lds_kernel, dim3(grid), dim3(1024), lds, (hipStream_t)stream, p, hot_ids, n_hot, col_hot, p.n_tasks / 32);
  if (p.n_fix > 0) {
    dmm::SpmmBatch b;
    memset(&b, 0, sizeof(b));
    b.n = 1; b.p[0] = p; b.p[0].fix_count = nullptr;
    hipLaunchKernelGGL(dmm::spmm_fixup_kernel<64>, dim3(p.n_fix, 1), dim3(256), 0, (hipStream_t)stream, b);
  }
  DMM_LAUNCH_CHECK("dmm_spmm_csr_bf16_hotlds");
  return 0;
}
}  // namespace dmm  (re-opened below)
namespace dmm {

template <int D, int VEC>
static int launch(const SpmmBatch& b, int blocks, hipStream_t s) {
  // Measured at 1M users x 500k items, d = 64 (profiles/README.md): fp32 operands are bound by the bytes that miss L2 (450-465 us in
  // every variant); bf16 operands are bound by the number of tasks in flight -- 4 gathers per lane in 58 VGPRs = 8 waves per SIMD x 8 tasks
  // per wave: 258 us, against 315 us with 8 gathers in 82 VGPRs (5 waves) and 520 us when 8 gathers are forced into 64 VGPRs (spills).
  static const int variant = [] { const char* e = getenv("DMM_SPMM_VARIANT"); return e ? atoi(e) : 0; }();  // tuning knob (read once)
  const bool deep = variant == 0 ? VEC == 4 : variant == 2;
  if (deep) hipLaunchKernelGGL((spmm_kernel<D, VEC, 8, 4>), dim3(blocks), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((spmm_kernel<D, VEC, 4, 8>), dim3(blocks), dim3(256), 0, s, b);
  int max_fix = 0;  // giant rows the main launch did not finish itself (no fix_count)
  for (int i = 0; i < b.n; ++i)
    if (!b.p[i].fix_count) max_fix = max_fix > b.p[i].n_fix ? max_fix : b.p[i].n_fix;
  if (max_fix > 0) hipLaunchKernelGGL(spmm_fixup_kernel<D>, dim3(max_fix, b.n), dim3(256), 0, s, b);
  DMM_LAUNCH_CHECK("dmm_spmm_csr");
  return 0;
}

}  // namespace dmm

extern "C" int dmm_spmm_tasks_per_block(int D) {
  if (D != 8 && D != 16 && D != 32 && D != 64 && D != 128 && D != 256) return -1;
  return 1024 / D;
}
extern "C" int dmm_spmm_tasks_per_block_bf16(int D) {
  if (D != 16 && D != 32 && D != 64 && D != 128 && D != 256) return -1;
  return 2048 / D;
}

static int spmm_fill_batch(const char* who, const dmm_spmm_problem* probs, int n_probs, int D, dmm::SpmmBatch& b, int& blocks, bool bf16 = false) {
  DMM_CHECK_ARG(probs && n_probs >= 1 && n_probs <= DMM_SPMM_MAX_BATCH, "%s: n_probs must be 1..%d", who, DMM_SPMM_MAX_BATCH);
  const int G = bf16 ? dmm_spmm_tasks_per_block_bf16(D) : dmm_spmm_tasks_per_block(D);
  DMM_CHECK_ARG(G > 0, "%s: D=%d unsupported (8,16,32,64,128,256)", who, D);
  memset(&b, 0, sizeof(b));
  b.n = n_probs;
  // XCD-contiguous placement of the single-task tail is opt-in (DMM_SPMM_REMAP=1, read once): measured neutral on the 4 MB TikTok operand
  // (X sits in every XCD's L2 + the Infinity Cache) and 20-30 % SLOWER on a 1M-user shard (contiguous slices of the row-sorted
  // task list give the XCDs unequal work: user rows are short, item rows long), so the dispatcher's round-robin order is kept.
  static const int remap_env = [] { const char* e = getenv("DMM_SPMM_REMAP"); return e ? (e[0] == '1') : 0; }();
  b.remap = remap_env;
  blocks = 0;
  for (int i = 0; i < n_probs; ++i) {
    const dmm_spmm_problem& p = probs[i];
    DMM_CHECK_ARG(p.tasks && p.col && p.val && p.x_lo, "%s: problem %d has a null operand", who, i);
    DMM_CHECK_ARG(p.n_tasks > 0 && p.n_tasks % G == 0, "%s: problem %d: n_tasks=%d is not a multiple of %d", who, i, p.n_tasks, G);
    DMM_CHECK_ARG(p.n_head_tasks >= 0 && p.n_head_tasks <= p.n_tasks && p.n_head_tasks % G == 0, "%s: problem %d: bad n_head_tasks", who, i);
    DMM_CHECK_ARG(p.y || p.y_pert || p.y16 || p.y_pert16, "%s: problem %d has no output", who, i);
    DMM_CHECK_ARG(p.n_partial_blocks >= 0 && p.n_partial_blocks * G <= p.n_head_tasks && p.n_fix >= 0, "%s: problem %d: bad giant-row fields", who, i);
    DMM_CHECK_ARG(p.n_partial_blocks == 0 || (p.partial && p.fix && p.n_fix > 0), "%s: problem %d: giant rows need partial + fix buffers", who, i);
    DMM_CHECK_ARG(!p.y_mean || ((p.y_pert || p.y_pert16) && p.m1 && p.m2), "%s: problem %d: y_mean needs y_pert, m1, m2", who, i);
    DMM_CHECK_ARG(p.x_hi || p.split >= (1 << 30), "%s: problem %d: x_hi missing", who, i);
    DMM_CHECK_ARG(!p.raw || (p.raw_from >= 0 && p.raw_from <= p.n_rows && p.n_rows > 0), "%s: problem %d: raw rows need 0 <= raw_from <= n_rows", who, i);
    b.p[i] = p;
    b.block_begin[i] = blocks;
    blocks += p.n_tasks / G;
  }
  for (int i = n_probs; i <= DMM_SPMM_MAX_BATCH; ++i) b.block_begin[i] = blocks;
  return 0;
}

extern "C" int dmm_spmm_csr_f32(const dmm_spmm_problem* probs, int n_probs, int D, void* stream) {
  dmm::SpmmBatch b;
  int blocks = 0;
  if (int rc = spmm_fill_batch("dmm_spmm_csr_f32", probs, n_probs, D, b, blocks)) return rc;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 8: return dmm::launch<8, 4>(b, blocks, s);
    case 16: return dmm::launch<16, 4>(b, blocks, s);
    case 32: return dmm::launch<32, 4>(b, blocks, s);
    case 64: return dmm::launch<64, 4>(b, blocks, s);
    case 128: return dmm::launch<128, 4>(b, blocks, s);
    case 256: return dmm::launch<256, 4>(b, blocks, s);
  }
  return DMM_ERR_INVALID_ARG;
}

// The same with the gathered operand X in bf16 ([rows, D] bf16, fp32 accumulation, fp32 epilogue operands and outputs; optional bf16
// copies of the outputs feed the next layer's launch).  The task list is planned for dmm_spmm_tasks_per_block_bf16(D) = 2048 / D groups.
extern "C" int dmm_spmm_csr_bf16(const dmm_spmm_problem* probs, int n_probs, int D, void* stream) {
  dmm::SpmmBatch b;
  int blocks = 0;
  if (int rc = spmm_fill_batch("dmm_spmm_csr_bf16", probs, n_probs, D, b, blocks, true)) return rc;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 16: return dmm::launch<16, 8>(b, blocks, s);
    case 32: return dmm::launch<32, 8>(b, blocks, s);
    case 64: return dmm::launch<64, 8>(b, blocks, s);
    case 128: return dmm::launch<128, 8>(b, blocks, s);
    case 256: return dmm::launch<256, 8>(b, blocks, s);
  }
  return DMM_ERR_INVALID_ARG;
}

// Second half of a user-sharded SpMM: the epilogue of the rows >= raw_from (the replicated item rows), run after the replicas'
// raw partial rows in `raw` have been summed.  Problems without `raw` are skipped.
extern "C" int dmm_spmm_epilogue_rows(const dmm_spmm_problem* probs, int n_probs, int D, void* stream) {
  dmm::SpmmBatch b;
  int blocks = 0;
  if (int rc = spmm_fill_batch("dmm_spmm_epilogue_rows", probs, n_probs, D, b, blocks)) return rc;
  int rows = 0;
  for (int i = 0; i < n_probs; ++i)
    if (probs[i].raw) rows = std::max(rows, probs[i].n_rows - probs[i].raw_from);
  if (rows == 0) return 0;
  const int G = 1024 / D;
  const dim3 grid(std::min<unsigned>(dmm::cdiv(rows, G), 2048u), n_probs);
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 8: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<8>, grid, dim3(256), 0, s, b); break;
    case 16: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<16>, grid, dim3(256), 0, s, b); break;
    case 32: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<32>, grid, dim3(256), 0, s, b); break;
    case 64: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<64>, grid, dim3(256), 0, s, b); break;
    case 128: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<128>, grid, dim3(256), 0, s, b); break;
    case 256: hipLaunchKernelGGL(dmm::spmm_epilogue_rows_kernel<256>, grid, dim3(256), 0, s, b); break;
    default: return DMM_ERR_INVALID_ARG;
  }
  DMM_LAUNCH_CHECK("dmm_spmm_epilogue_rows");
  return 0;
}
