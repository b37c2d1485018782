// Rebuild of a modality graph on the device: from the per-user item lists that the reverse chain's top-k wrote
// (Main.py:215-230) to the normalised bipartite adjacency D^-1/2 (A + I) D^-1/2 in CSR (DataHandler.makeTorchAdj /
// normalizeAdj, DataHandler.py:54-93, on the edges of Main.py:232-253) AND the SpMM work list of spmm.hip -- bit for bit what
// diffmm_amd/graph.py builds on the host (rows sorted by column, values rounded from the same fp64 products, the same task
// order), with no host round trip of the edges and one 32-byte read-back (the list sizes).
//
//   1. gb_count_items     item degrees (integer atomics: exact whatever the order)
//   2. gb_scan_partial / gb_scan_blocks / gb_scan_emit
//                         three-phase multi-channel exclusive scan over the N = U + I rows: channel 0 = row length (-> rowptr),
//                         one channel per task class (giant segments, split rows by power-of-two run size, single-task rows;
//                         the planner's order is class-major, row-ascending inside a class) and one counting the giant rows;
//                         the last phase also writes every row's tasks
//   3. gb_fill            user rows (self loop, then the user's items in top-k order) and the unsorted contents of the item rows
//   4. gb_finish_item_rows  every item row sorted by user id (rank by counting in LDS; a bitmap pass for rows longer than 256 entries),
//                         self loop last, values
// All index arithmetic is 32-bit: nnz = 2 E + N < 2^31 is checked by the host wrapper.
#include "common.h"

namespace dmm {

constexpr int GB_MAXCH = 12;      // row length + up to 10 task classes + giant-row count
constexpr int GB_SCAN_BLOCK = 1024;
constexpr int GB_SORT_CAP = 256;  // item-row entries ranked by counting in LDS (O(L^2)); longer rows take the bitmap pass (O(U/32 + L))

struct GbShape {
  int U, I, N, E, G, logG, task_len, seg_len, nch;  // nch = 1 + (logG + 2) + 1
  // user-sharded block: U = the replica's own users, item i keeps its self loop here iff i % self_mod == self_rank, and the
  // item degrees that scale the values are the GLOBAL ones (`gdeg`, summed over the replicas), not the local row lengths (`cnt`)
  int self_mod, self_rank;
};
__device__ __forceinline__ int gb_own(const GbShape& S, int it) { return (it % S.self_mod) == S.self_rank ? 1 : 0; }

// meta (device, int32[8]): 0 n_tasks, 1 n_head_tasks, 2 n_partial_blocks, 3 n_fix, 4 nnz
__device__ __forceinline__ int gb_row_len(const GbShape& S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ cnt, int r) {
  return r < S.U ? user_ptr[r + 1] - user_ptr[r] + 1 : cnt[r - S.U] + gb_own(S, r - S.U);
}

// class of a row and the number of task slots it reserves: 0 = giant (nseg * G), 1 .. logG = split rows with run size G .. 2,
// logG + 1 = single task
__device__ __forceinline__ void gb_classify(const GbShape& S, int len, int& cls, int& w) {
  if (len > S.seg_len) {
    cls = 0;
    w = ((len + S.seg_len - 1) / S.seg_len) * S.G;
    return;
  }
  int k = (len + S.task_len - 1) / S.task_len;
  k = k < 1 ? 1 : (k > S.G ? S.G : k);
  if (k == 1) {
    cls = S.logG + 1;
    w = 1;
    return;
  }
  int lg = 32 - __clz(k - 1);  // ceil(log2 k)
  cls = 1 + (S.logG - lg);
  w = 1 << lg;
}

__device__ __forceinline__ void gb_channels(const GbShape& S, int len, int cls, int w, int (&ch)[GB_MAXCH]) {
#pragma unroll
  for (int c = 0; c < GB_MAXCH; ++c) ch[c] = 0;
  ch[0] = len;
#pragma unroll
  for (int c = 0; c < GB_MAXCH - 2; ++c)
    if (c == cls) ch[1 + c] = w;
  if (cls == 0) {
#pragma unroll
    for (int c = 0; c < GB_MAXCH; ++c)
      if (c == S.nch - 1) ch[c] = 1;
  }
}

// exclusive scan of one int per thread over a 1024-thread workgroup; returns the exclusive prefix, *total = workgroup sum
__device__ __forceinline__ int gb_block_scan(int v, int* total, int* lds /*[17]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // lds reuse between calls
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < GB_SCAN_BLOCK / 64; ++i) {
      const int t = lds[i];
      lds[i] = run;
      run += t;
    }
    lds[16] = run;
  }
  __syncthreads();
  *total = lds[16];
  return inc - v + lds[wave];
}

__global__ __launch_bounds__(256) void gb_count_items(const int32_t* __restrict__ item, int E, int I, int32_t* __restrict__ cnt) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256) {
    const int it = item[e];
    if ((unsigned)it < (unsigned)I) atomicAdd(&cnt[it], 1);
  }
}

// The same with a workgroup-private histogram in LDS (I <= GB_LDS_ITEMS): a popular item of a rebuilt graph is chosen by thousands of
// users, and that many atomics on ONE address serialise in L2 (~100 us at TikTok size); here a hot item costs one global atomic per
// workgroup.  Each workgroup takes a contiguous chunk of GB_CHUNK edges.
constexpr int GB_CHUNK = 2048;
constexpr int GB_LDS_ITEMS = 18 * 1024;  // 2 x 72 KiB of LDS in gb_fill_lds

__global__ __launch_bounds__(256) void gb_count_items_lds(const int32_t* __restrict__ item, int E, int I, int32_t* __restrict__ cnt) {
  extern __shared__ int gb_h[];
  for (int b = threadIdx.x; b < I; b += 256) gb_h[b] = 0;
  __syncthreads();
  const int e0 = blockIdx.x * GB_CHUNK, e1 = min(E, e0 + GB_CHUNK);
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    const int it = item[e];
    if ((unsigned)it < (unsigned)I) atomicAdd(&gb_h[it], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < I; b += 256) {
    const int c = gb_h[b];
    if (c) atomicAdd(&cnt[b], c);
  }
}

// phase 1: per-workgroup channel sums
__global__ __launch_bounds__(GB_SCAN_BLOCK) void gb_scan_partial(GbShape S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ cnt,
                                                                int32_t* __restrict__ blocksum /*[nch][nb]*/) {
  __shared__ int lds[17];
  const int r = blockIdx.x * GB_SCAN_BLOCK + threadIdx.x;
  int ch[GB_MAXCH];
  if (r < S.N) {
    const int len = gb_row_len(S, user_ptr, cnt, r);
    int cls, w;
    gb_classify(S, len, cls, w);
    gb_channels(S, len, cls, w, ch);
  } else {
#pragma unroll
    for (int c = 0; c < GB_MAXCH; ++c) ch[c] = 0;
  }
#pragma unroll
  for (int c = 0; c < GB_MAXCH; ++c) {
    if (c < S.nch) {
      int tot;
      (void)gb_block_scan(ch[c], &tot, lds);
      if (threadIdx.x == 0) blocksum[c * gridDim.x + blockIdx.x] = tot;
    }
  }
}

// phase 2 (one workgroup): exclusive scan of the workgroup sums per channel, region bases of the task list, pads, meta
__global__ __launch_bounds__(GB_SCAN_BLOCK) void gb_scan_blocks(GbShape S, int nb, int32_t* __restrict__ blocksum, int32_t* __restrict__ base /*[GB_MAXCH]*/,
                                                               int4* __restrict__ tasks, int32_t* __restrict__ meta, int32_t* __restrict__ rowptr) {
  __shared__ int lds[17];
  __shared__ int total[GB_MAXCH];
  for (int c = 0; c < S.nch; ++c) {
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += GB_SCAN_BLOCK) {
      const int b = b0 + threadIdx.x;
      const int v = b < nb ? blocksum[c * nb + b] : 0;
      int tot;
      const int ex = gb_block_scan(v, &tot, lds);
      if (b < nb) blocksum[c * nb + b] = carry + ex;
      carry += tot;
    }
    if (threadIdx.x == 0) total[c] = carry;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // task list: [giant segments | runs of G | runs of G/2 | ... | runs of 2 | pad to G | single-task rows | pad to G]
    int off = 0;
    base[0] = 0;
    off += total[1];  // giant slots (nseg * G per row)
    int multi = 0;
    for (int c = 1; c <= S.logG; ++c) {
      base[c] = off;
      off += total[1 + c];
      multi += total[1 + c];
    }
    const int pad_m = multi ? (S.G - multi % S.G) % S.G : 0;
    const int head = off + pad_m;
    base[S.logG + 1] = head;
    const int singles = total[1 + S.logG + 1];
    const int pad_s = singles ? (S.G - singles % S.G) % S.G : 0;
    const int n_tasks = head + singles + pad_s;
    base[GB_MAXCH - 1] = n_tasks;
    meta[0] = n_tasks;
    meta[1] = head;
    meta[2] = total[1] / S.G;     // segments of giant rows = partial workgroups
    meta[3] = total[S.nch - 1];   // giant rows = fix-up entries
    meta[4] = total[0];           // nnz
    meta[5] = off;                // end of the split-row region before its pad
    meta[6] = head + singles;     // end of the single-task region before its pad
    rowptr[S.N] = total[0];
  }
  __syncthreads();
  const int4 padt = make_int4(-1, 0, 0, 1);
  const int m5 = meta[5], m1 = meta[1], m6 = meta[6], m0 = meta[0];
  for (int i = m5 + threadIdx.x; i < m1; i += GB_SCAN_BLOCK) tasks[i] = padt;
  for (int i = m6 + threadIdx.x; i < m0; i += GB_SCAN_BLOCK) tasks[i] = padt;
}

// phase 3: final prefixes; rowptr; every row writes its tasks (and its fix-up entry if it is a giant row)
__global__ __launch_bounds__(GB_SCAN_BLOCK) void gb_scan_emit(GbShape S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ cnt,
                                                             const int32_t* __restrict__ blockoff, const int32_t* __restrict__ base,
                                                             int32_t* __restrict__ rowptr, int4* __restrict__ tasks, int4* __restrict__ fix) {
  __shared__ int lds[17];
  const int r = blockIdx.x * GB_SCAN_BLOCK + threadIdx.x;
  int ch[GB_MAXCH];
  int len = 0, cls = -1, w = 0;
  if (r < S.N) {
    len = gb_row_len(S, user_ptr, cnt, r);
    gb_classify(S, len, cls, w);
    gb_channels(S, len, cls, w, ch);
  } else {
#pragma unroll
    for (int c = 0; c < GB_MAXCH; ++c) ch[c] = 0;
  }
  int rp = 0, toff = 0, grank = 0;
#pragma unroll
  for (int c = 0; c < GB_MAXCH; ++c) {
    if (c < S.nch) {
      int tot;
      const int ex = gb_block_scan(ch[c], &tot, lds) + blockoff[c * gridDim.x + blockIdx.x];
      if (c == 0) rp = ex;
      if (c == 1 + cls) toff = ex;
      if (c == S.nch - 1) grank = ex;
    }
  }
  if (r >= S.N) return;
  rowptr[r] = rp;
  const int4 padt = make_int4(-1, 0, 0, 1);
  if (cls == S.logG + 1) {
    tasks[base[cls] + toff] = make_int4(r, rp, rp + len, 1);
  } else if (cls > 0) {
    const int kp = w, chunk = (len + kp - 1) / kp, nlive = (len + chunk - 1) / chunk;
    int4* t = tasks + base[cls] + toff;
    for (int j = 0; j < kp; ++j) {
      const int beg = rp + j * chunk, end = min(beg + chunk, rp + len);
      t[j] = beg < end ? make_int4(r, beg, end, (j << 16) | nlive) : padt;
    }
  } else {
    const int nseg = w / S.G, first = toff / S.G;
    fix[grank] = make_int4(r, first, nseg, 0);
    for (int s = 0; s < nseg; ++s) {
      const int sbeg = rp + s * S.seg_len, send = min(sbeg + S.seg_len, rp + len);
      const int chunk = max(1, (send - sbeg + S.G - 1) / S.G), nlive = (send - sbeg + chunk - 1) / chunk;
      int4* t = tasks + (first + s) * S.G;
      for (int g = 0; g < S.G; ++g) {
        const int beg = sbeg + g * chunk, end = min(beg + chunk, send);
        t[g] = beg < end ? make_int4(r, beg, end, (g << 16) | nlive) : padt;
      }
    }
  }
}

// value of entry (row, col): fp64 product of the two D^-1/2 factors, rounded once -- ((dinv[row] * 1.0) * dinv[col]).float()
__device__ __forceinline__ float gb_val(const double* __restrict__ dinv, int len_row, int len_col) { return (float)((dinv[len_row] * 1.0) * dinv[len_col]); }

__global__ __launch_bounds__(256) void gb_fill(GbShape S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ edge_user,
                                               const int32_t* __restrict__ item, const int32_t* __restrict__ cnt, const int32_t* __restrict__ rowptr,
                                               const double* __restrict__ dinv, int32_t* __restrict__ fill, int32_t* __restrict__ col, float* __restrict__ val,
                                               const int32_t* __restrict__ gdeg) {
  const int stride = gridDim.x * 256;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < S.E; e += stride) {
    const int u = edge_user[e], it = item[e];
    const int lu = user_ptr[u + 1] - user_ptr[u] + 1, li = gdeg[it] + 1;
    const int p = rowptr[u] + 1 + (e - user_ptr[u]);  // a user's items keep their (ascending) top-k order, after the self loop
    col[p] = S.U + it;
    val[p] = gb_val(dinv, lu, li);
    const int q = rowptr[S.U + it] + atomicAdd(&fill[it], 1);  // unsorted; gb_finish_item_rows orders the row
    col[q] = u;
  }
  for (int u = blockIdx.x * 256 + threadIdx.x; u < S.U; u += stride) {
    const int lu = user_ptr[u + 1] - user_ptr[u] + 1, p = rowptr[u];
    col[p] = u;
    val[p] = gb_val(dinv, lu, lu);
  }
}

// gb_fill with workgroup-private slot allocation (I <= GB_LDS_ITEMS): an edge takes its rank inside the workgroup's chunk from an LDS
// counter; the chunk's block of slots in the item row comes from ONE global atomic per (workgroup, item).  Slot order inside an item row
// is arbitrary either way (gb_finish_item_rows orders the row).
__global__ __launch_bounds__(256) void gb_fill_lds(GbShape S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ edge_user,
                                                   const int32_t* __restrict__ item, const int32_t* __restrict__ cnt, const int32_t* __restrict__ rowptr,
                                                   const double* __restrict__ dinv, int32_t* __restrict__ fill, int32_t* __restrict__ col,
                                                   float* __restrict__ val, const int32_t* __restrict__ gdeg) {
  extern __shared__ int gb_h[];  // [I] local counts, then [I] bases
  int* hb = gb_h + S.I;
  for (int b = threadIdx.x; b < S.I; b += 256) gb_h[b] = 0;
  __syncthreads();
  constexpr int PER = GB_CHUNK / 256;
  const int e0 = blockIdx.x * GB_CHUNK, e1 = min(S.E, e0 + GB_CHUNK);
  int lr[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int e = e0 + q * 256 + threadIdx.x;
    lr[q] = e < e1 ? atomicAdd(&gb_h[item[e]], 1) : 0;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < S.I; b += 256) {
    const int c = gb_h[b];
    if (c) hb[b] = atomicAdd(&fill[b], c);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int e = e0 + q * 256 + threadIdx.x;
    if (e < e1) {
      const int u = edge_user[e], it = item[e];
      const int lu = user_ptr[u + 1] - user_ptr[u] + 1, li = gdeg[it] + 1;
      const int p = rowptr[u] + 1 + (e - user_ptr[u]);
      col[p] = S.U + it;
      val[p] = gb_val(dinv, lu, li);
      col[rowptr[S.U + it] + hb[it] + lr[q]] = u;
    }
  }
  for (int u = blockIdx.x * 256 + threadIdx.x; u < S.U; u += gridDim.x * 256) {
    const int lu = user_ptr[u + 1] - user_ptr[u] + 1, p = rowptr[u];
    col[p] = u;
    val[p] = gb_val(dinv, lu, lu);
  }
}

// One workgroup per item row: users ascending, self loop last.  Users of a row are distinct, so "number of smaller entries" is a
// permutation.  Rows longer than GB_SORT_CAP take the bitmap path (one U-bit map per workgroup in `bitmaps`).
__global__ __launch_bounds__(256) void gb_finish_item_rows(GbShape S, const int32_t* __restrict__ user_ptr, const int32_t* __restrict__ cnt,
                                                           const int32_t* __restrict__ rowptr, const double* __restrict__ dinv,
                                                           int32_t* __restrict__ col, float* __restrict__ val, uint32_t* __restrict__ bitmaps, int long_rows,
                                                           const int32_t* __restrict__ gdeg) {
  __shared__ int ent[GB_SORT_CAP];
  __shared__ int wsum[257];
  const int words = (S.U + 31) >> 5;
  for (int it = blockIdx.x; it < S.I; it += gridDim.x) {
    const int L = cnt[it], rp = rowptr[S.U + it], li = gdeg[it] + 1;
    if ((L > GB_SORT_CAP) != (long_rows != 0)) continue;
    if (!long_rows) {
      for (int i = threadIdx.x; i < L; i += 256) ent[i] = col[rp + i];
      __syncthreads();
      for (int i = threadIdx.x; i < L; i += 256) {
        const int x = ent[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) rank += ent[j] < x;
        col[rp + rank] = x;
        val[rp + rank] = gb_val(dinv, li, user_ptr[x + 1] - user_ptr[x] + 1);
      }
      __syncthreads();
    } else {
      uint32_t* bm = bitmaps + (size_t)blockIdx.x * words;
      for (int i = threadIdx.x; i < words; i += 256) bm[i] = 0u;
      __syncthreads();
      for (int i = threadIdx.x; i < L; i += 256) {
        const int x = col[rp + i];
        atomicOr(&bm[x >> 5], 1u << (x & 31));
      }
      __syncthreads();
      const int per = (words + 255) / 256, w0 = threadIdx.x * per, w1 = min(words, w0 + per);
      int c = 0;
      for (int i = w0; i < w1; ++i) c += __popc(bm[i]);
      wsum[threadIdx.x + 1] = c;
      __syncthreads();
      if (threadIdx.x == 0) {
        wsum[0] = 0;
        for (int i = 1; i <= 256; ++i) wsum[i] += wsum[i - 1];
      }
      __syncthreads();
      int pos = rp + wsum[threadIdx.x];
      for (int i = w0; i < w1; ++i) {
        uint32_t b = bm[i];
        while (b) {
          const int x = (i << 5) + __ffs(b) - 1;
          b &= b - 1;
          col[pos] = x;
          val[pos] = gb_val(dinv, li, user_ptr[x + 1] - user_ptr[x] + 1);
          ++pos;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && gb_own(S, it)) {
      col[rp + L] = S.U + it;
      val[rp + L] = gb_val(dinv, li, li);
    }
  }
}

constexpr int GB_LONG_GROUPS = 256;

}  // namespace dmm
using namespace dmm;

static int gb_shape(GbShape& S, int U, int I, int64_t E, int D, int task_len) {
  DMM_CHECK_ARG(U > 0 && I > 0 && E >= 0 && task_len >= 1, "dmm_graph_build: bad sizes");
  DMM_CHECK_ARG(D >= 4 && D <= 512 && 1024 % D == 0 && D % 4 == 0, "dmm_graph_build: D=%d unsupported (1024 %% D == 0, D %% 4 == 0)", D);
  DMM_CHECK_ARG(2 * E + U + I < (int64_t)0x7fffffff, "dmm_graph_build: nnz does not fit 32-bit indices");
  S.U = U; S.I = I; S.N = U + I; S.E = (int)E; S.G = 1024 / D; S.task_len = task_len;
  S.self_mod = 1; S.self_rank = 0;
  S.logG = 0;
  while ((1 << S.logG) < S.G) ++S.logG;
  DMM_CHECK_ARG((1 << S.logG) == S.G, "dmm_graph_build: 1024 / D must be a power of two");
  S.seg_len = 2 * S.G * task_len;
  S.nch = 1 + (S.logG + 2) + 1;
  DMM_CHECK_ARG(S.nch <= GB_MAXCH, "dmm_graph_build: too many task classes");
  return 0;
}

extern "C" int dmm_graph_build_scratch_bytes(int U, int I, int64_t E, int D, int64_t* bytes) {
  GbShape S;
  if (int rc = gb_shape(S, U, I, E, D, 32)) return rc;
  DMM_CHECK_ARG(bytes, "dmm_graph_build_scratch_bytes: null output");
  const int64_t nb = (S.N + GB_SCAN_BLOCK - 1) / GB_SCAN_BLOCK;
  const int64_t words = (U + 31) / 32;
  *bytes = 4 * ((int64_t)2 * I + GB_MAXCH * nb + GB_MAXCH + GB_LONG_GROUPS * words) + 256;
  return 0;
}

extern "C" int dmm_graph_build_task_capacity(int U, int I, int64_t E, int D, int task_len, int64_t* tasks, int64_t* fix) {
  GbShape S;
  if (int rc = gb_shape(S, U, I, E, D, task_len)) return rc;
  DMM_CHECK_ARG(tasks && fix, "dmm_graph_build_task_capacity: null output");
  const int64_t nnz = 2 * E + S.N;
  // split rows reserve < 2 * len / task_len + 2 slots, single rows 1, giant rows < len * G / seg_len + G; two pads of < G
  *tasks = 2 * (nnz / task_len) + 2 * (int64_t)S.N + nnz / (2 * task_len) + (int64_t)S.G * (nnz / S.seg_len + 4) + 4 * S.G;
  *fix = nnz / S.seg_len + 1;
  return 0;
}

extern "C" int dmm_graph_build(const int32_t* user_ptr, const int32_t* edge_user, const int32_t* item_idx, int U, int I, int64_t E, int D, int task_len,
                               const double* dinv_table, int32_t* rowptr, int32_t* col, float* val, void* tasks, void* fix, int32_t* meta,
                               void* scratch, void* stream) {
  return dmm_graph_build_shard(user_ptr, edge_user, item_idx, U, I, E, D, task_len, dinv_table, nullptr, 1, 0, rowptr, col, val, tasks, fix, meta, scratch,
                               stream);
}

extern "C" int dmm_count_items(const int32_t* item_idx, int64_t E, int I, int32_t* count, void* stream) {
  DMM_CHECK_ARG(count && I > 0 && E >= 0 && E < (int64_t)0x7fffffff && (item_idx || E == 0), "dmm_count_items: bad arguments");
  if (E == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (I <= GB_LDS_ITEMS) {
    static bool once = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gb_count_items_lds), hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS_ITEMS * 4);
      return true;
    }();
    (void)once;
    hipLaunchKernelGGL(gb_count_items_lds, dim3((unsigned)((E + GB_CHUNK - 1) / GB_CHUNK)), dim3(256), (size_t)I * 4, s, item_idx, (int)E, I, count);
  } else {
    hipLaunchKernelGGL(gb_count_items, dim3((unsigned)std::min<int64_t>(4096, (E + 255) / 256)), dim3(256), 0, s, item_idx, (int)E, I, count);
  }
  DMM_LAUNCH_CHECK("dmm_count_items");
  return 0;
}

extern "C" int dmm_graph_build_shard(const int32_t* user_ptr, const int32_t* edge_user, const int32_t* item_idx, int U, int I, int64_t E, int D,
                                     int task_len, const double* dinv_table, const int32_t* item_count_global, int self_mod, int self_rank,
                                     int32_t* rowptr, int32_t* col, float* val, void* tasks, void* fix, int32_t* meta, void* scratch, void* stream) {
  GbShape S;
  if (int rc = gb_shape(S, U, I, E, D, task_len)) return rc;
  DMM_CHECK_ARG(self_mod >= 1 && self_rank >= 0 && self_rank < self_mod, "dmm_graph_build_shard: need 0 <= self_rank < self_mod");
  S.self_mod = self_mod; S.self_rank = self_rank;
  DMM_CHECK_ARG(user_ptr && edge_user && (item_idx || E == 0) && dinv_table && rowptr && col && val && tasks && fix && meta && scratch,
                "dmm_graph_build: null operand");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (S.N + GB_SCAN_BLOCK - 1) / GB_SCAN_BLOCK;
  int32_t* cnt = (int32_t*)scratch;
  int32_t* fill = cnt + I;
  int32_t* blocksum = fill + I;
  int32_t* base = blocksum + (size_t)GB_MAXCH * nb;
  uint32_t* bitmaps = (uint32_t*)(base + GB_MAXCH);
  const int32_t* gdeg = item_count_global ? item_count_global : cnt;  // unsharded: the local counts ARE the degrees
  if (hipMemsetAsync(cnt, 0, sizeof(int32_t) * 2 * (size_t)I, s) != hipSuccess) {
    set_error("dmm_graph_build: hipMemsetAsync failed");
    return 2;
  }
  const int eg = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (std::max<int64_t>(E, U) + 255) / 256));
  const bool lds_path = I <= GB_LDS_ITEMS && E > 0;
  const int cg = (int)((E + GB_CHUNK - 1) / GB_CHUNK);
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gb_count_items_lds), hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS_ITEMS * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gb_fill_lds), hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS_ITEMS * 8);
    return true;
  }();
  (void)once;
  if (lds_path) hipLaunchKernelGGL(gb_count_items_lds, dim3(cg), dim3(256), (size_t)I * 4, s, item_idx, S.E, I, cnt);
  else if (E > 0) hipLaunchKernelGGL(gb_count_items, dim3(eg), dim3(256), 0, s, item_idx, S.E, I, cnt);
  hipLaunchKernelGGL(gb_scan_partial, dim3(nb), dim3(GB_SCAN_BLOCK), 0, s, S, user_ptr, (const int32_t*)cnt, blocksum);
  hipLaunchKernelGGL(gb_scan_blocks, dim3(1), dim3(GB_SCAN_BLOCK), 0, s, S, nb, blocksum, base, (int4*)tasks, meta, rowptr);
  hipLaunchKernelGGL(gb_scan_emit, dim3(nb), dim3(GB_SCAN_BLOCK), 0, s, S, user_ptr, (const int32_t*)cnt, (const int32_t*)blocksum, (const int32_t*)base,
                     rowptr, (int4*)tasks, (int4*)fix);
  if (lds_path)
    hipLaunchKernelGGL(gb_fill_lds, dim3(cg), dim3(256), (size_t)I * 8, s, S, user_ptr, edge_user, item_idx, (const int32_t*)cnt, (const int32_t*)rowptr,
                       dinv_table, fill, col, val, gdeg);
  else
    hipLaunchKernelGGL(gb_fill, dim3(eg), dim3(256), 0, s, S, user_ptr, edge_user, item_idx, (const int32_t*)cnt, (const int32_t*)rowptr, dinv_table, fill,
                       col, val, gdeg);
  hipLaunchKernelGGL(gb_finish_item_rows, dim3(std::min(I, 65535)), dim3(256), 0, s, S, user_ptr, (const int32_t*)cnt, (const int32_t*)rowptr, dinv_table,
                     col, val, bitmaps, 0, gdeg);
  hipLaunchKernelGGL(gb_finish_item_rows, dim3(GB_LONG_GROUPS), dim3(256), 0, s, S, user_ptr, (const int32_t*)cnt, (const int32_t*)rowptr, dinv_table, col,
                     val, bitmaps, 1, gdeg);
  DMM_LAUNCH_CHECK("dmm_graph_build");
  return 0;
}
