/* libdiffmm_hip -- C ABI of the MI355X (gfx950) kernels behind the DiffMM training hot path.
 *
 * The reference (sun2ot/DiffMM) has no native layer: its hot path is a sequence of PyTorch op call sites.
 * Each entry point below replaces the call sites it cites (file:line in /root/reference); the Python
 * surface in diffmm_amd/ (Model, Denoise, GaussianDiffusion, InfoNCE, bpr_loss, Coach ...) binds them
 * through ctypes -- see INTEGRATION.md for the stubs a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator) unless marked "host";
 *   - every launch is asynchronous on the given hipStream_t (passed as void*); no hidden allocation,
 *     no hidden synchronisation, no global mutable state (the last-error string is thread-local, like errno);
 *   - return value 0 = ok; >0 = hipError_t of a failed launch; <0 = DMM_ERR_*; text via
 *     dmm_last_error_string();
 *   - matrices are row-major fp32 unless stated; indices are int32.
 */
#ifndef DIFFMM_HIP_H
#define DIFFMM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DMM_VERSION 400
#define DMM_ERR_INVALID_ARG (-1)
#define DMM_ERR_UNSUPPORTED (-2)
#define DMM_SPMM_MAX_BATCH 4

int dmm_version(void);
const char* dmm_last_error_string(void);

/* ---------------------------------------------------------------------------------------------------
 * CSR SpMM with fused epilogues.  Replaces torch.sparse.mm(adj, X) and its backward (A symmetric):
 * Model.py:90,93,105,111,114,123,130 and Main.py:319; the epilogues absorb Model.py:116-131 (scale /
 * residual adds), Main.py:320-321 (sign-aligned noise) and Main.py:325 (mean over the 3 CL layers).
 *   y      = alpha * (A X)[row] + beta * c1[row] + gamma * c2[row]
 *   y_pert = y + sign(y) * noise[row] / max(||noise[row]||, 1e-12) * noise_degree
 *   y_mean = (y_pert + m1[row] + m2[row]) / 3
 * X row r is read from x_lo[r] when r < split, else x_hi[r - split] (lets [u_embs ; feats] stay two tensors).
 * `tasks` is the int4 work list {row, nz_begin, nz_end, run_pos<<16 | run_len} produced by the planner
 * (diffmm_amd/graph.py: plan_tasks); n_tasks must be a multiple of dmm_spmm_tasks_per_block(D).  The planner puts the
 * tasks of split (heavy) rows first: those n_head_tasks keep the hardware's round-robin workgroup->XCD dispatch (they hold
 * most of the non-zeros and must be spread), the single-task tail is remapped so each XCD walks a contiguous row range.
 * Rows beyond one workgroup's worth of tasks ("giant": a popular item of a rebuilt graph) are cut into segments, the first
 * n_partial_blocks workgroups of the list; they park raw partial rows in `partial`, and a second tiny launch adds them in segment
 * order and applies the epilogue (deterministic, no atomics).
 * noise == NULL with y_pert != NULL draws uniform noise in-kernel: Philox4x32-10(seed, offset + row*D/4 + lane). */
typedef struct dmm_spmm_problem {
  const void* tasks;
  const int32_t* col;
  const float* val;
  const float* x_lo;
  const float* x_hi;
  float* y;
  const float* c1;
  const float* c2;
  float* y_pert;
  const float* noise;
  float* y_mean;
  const float* m1;
  const float* m2;
  uint64_t seed;
  uint64_t offset;
  float* partial;       /* [n_partial_blocks, D] scratch for the segments of giant rows (NULL when there are none) */
  const void* fix;      /* int4 {row, first_slot, n_slots, 0} per giant row */
  int32_t n_partial_blocks, n_fix;
  int32_t n_rows;  /* rows of the adjacency (sizes the XCD-placement heuristic) */
  int32_t n_tasks;
  int32_t n_head_tasks; /* leading tasks (split heavy rows, multiple of tasks_per_block) dispatched round-robin over the XCDs */
  int32_t split;
  float alpha, beta, gamma;
  float noise_degree;
  /* User-sharded layout (one replica owns a user range, items replicated: SURVEY 8e).  When raw != NULL, rows >= raw_from (the item
   * rows of the local block [own users ; all items]) are only a replica's PARTIAL sums: the launch parks their raw accumulators in
   * raw [n_rows - raw_from, D] and applies no epilogue to them; the caller sums `raw` over the replicas (one all-reduce per launch)
   * and then calls dmm_spmm_epilogue_rows on the same problem array.  Rows < raw_from (own users) are complete and finished in place. */
  float* raw;
  int32_t raw_from;
  /* In-kernel Philox draws are keyed by the row's GLOBAL id: row + (row < rng_row_split ? rng_row_add_lo : rng_row_add_hi), so a
   * replica draws what the unsharded launch draws for that row (all zero = local ids, the unsharded case). */
  int32_t rng_row_split, rng_row_add_lo, rng_row_add_hi;
  /* optional bf16 copies of y / y_pert ([n_rows, D] bf16): the operand of the next layer's dmm_spmm_csr_bf16 launch, written by the
   * producing launch instead of a separate cast pass */
  void* y16;
  void* y_pert16;
  /* Giant rows finished inside the main launch (optional; NULL = the second launch described above).  int32 [n_fix], ZERO on entry and zero
   * again on return: every segment workgroup of a giant row parks its partial row, publishes it (agent-scope release) and adds 1 to the row's
   * counter; the workgroup whose add comes last (agent-scope acquire) sums the row's segments IN SEGMENT ORDER -- the same order and grouping
   * as the second launch, so the result is bit-identical to it and reproducible whichever workgroup is last --, applies the epilogue and resets
   * the counter.  One buffer per concurrent use of an adjacency (like `partial`); a launch aborted mid-way leaves it dirty (re-zero it). */
  int32_t* fix_count;
} dmm_spmm_problem;

int dmm_spmm_tasks_per_block(int D);
int dmm_spmm_csr_f32(const dmm_spmm_problem* probs /*host*/, int n_probs, int D, void* stream);
/* bf16-storage variant (BASELINE.json configs 2-5, SURVEY 8d "bf16 storage / fp32 accumulate"): x_lo / x_hi point to bf16 rows
 * ([rows, D] bf16, 16-byte aligned), the gather moves half the bytes and one wave carries twice the tasks; accumulation, epilogue
 * operands (c1, c2, noise, m1, m2) and outputs stay fp32, y16 / y_pert16 add bf16 copies.  The task list must be planned for
 * dmm_spmm_tasks_per_block_bf16(D) = 2048 / D groups per workgroup (diffmm_amd/graph.py: plan width D / 2).  Same fix-up pass,
 * same determinism; the in-kernel Philox draws are the ones dmm_spmm_csr_f32 makes. */
int dmm_spmm_tasks_per_block_bf16(int D);
int dmm_spmm_csr_bf16(const dmm_spmm_problem* probs /*host*/, int n_probs, int D, void* stream);
/* EXPERIMENT (not on the product path; tools/spmm_scale.py --variants bf16_x_hot): dmm_spmm_csr_bf16 for ONE plain problem y = A x at D = 64 with the
 * n_hot (<= 896) most-referenced X rows staged in every persistent workgroup's LDS -- north_star's "LDS-staged embedding tiles" in the only form
 * that does not multiply the Y traffic at this density.  hot_ids [n_hot]: the staged rows; col_hot [nnz]: the column array with row hot_ids[s]
 * written as -1 - s; grid: persistent workgroups (1024 threads, one per CU).  Giant rows: partial + fix as usual (second launch). */
int dmm_spmm_csr_bf16_hotlds(const dmm_spmm_problem* prob /*host*/, const int32_t* hot_ids, int n_hot, const int32_t* col_hot, int grid, void* stream);
/* epilogue of the rows >= raw_from from the (all-reduced) raw partial rows; problems with raw == NULL are skipped */
int dmm_spmm_epilogue_rows(const dmm_spmm_problem* probs /*host*/, int n_probs, int D, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Dense GEMM on the matrix cores (bf16 MFMA 32x32x16, fp32 accumulate; operands stay fp32 in HBM and are
 * rounded while staged into LDS).  precision = 1: one bf16 product; precision = 3: hi/lo split, three
 * products (~fp32 accuracy) -- the mode the parity tests use.
 *   C[m,n] = post( act( alpha * sum_k A(m,k) B(k,n) + bias[n] ) )     post(v) = e1 * v + e2 * R[m,n]
 * A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]  (any of NN / NT / TN by strides).
 * split_k > 1: K is cut in split_k slices added into C with fp32 atomics (C must be pre-zeroed or hold the
 * value to accumulate into; only act = NONE, R = NULL; the bias is added by slice 0).
 * Replaces nn.Linear / torch.mm call sites: Model.py:48,51,57,84-85,103 (modality projection), :205-208
 * (gate), :211-215 (Denoise MLP), :375 (posterior mean as epilogue), :416-417 and their backward. */
enum { DMM_ACT_NONE = 0, DMM_ACT_TANH = 1, DMM_ACT_SIGMOID = 2 };
typedef struct dmm_gemm_args {
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t ldc;
  const float* bias;  /* [N] or NULL */
  const float* R; int64_t ldr; /* residual or NULL */
  int32_t M, N, K;
  float alpha, e1, e2;
  int32_t act;
  int32_t split_k;
  int32_t precision; /* 1 or 3 */
  int32_t accumulate; /* 1: C += result (plain read-modify-write, split_k == 1) */
} dmm_gemm_args;
int dmm_gemm(const dmm_gemm_args* a /*host*/, void* stream);
/* up to 4 problems of one operand layout (NN / NT / TN) and precision in one launch; M, N, K, split_k and the epilogues are per problem
 * (the modalities' projections Linear_m(E_m) and their weight gradients: Model.py:84-85,103) */
int dmm_gemm_batched(const dmm_gemm_args* a /*host array*/, int n, void* stream);

/* bf16 NT GEMM of the reverse diffusion chain (GaussianDiffusion.generate_view, Model.py:300-322: T x Denoise.forward +
 * posterior mean): A [M,K] and B [N,K] are bf16 row-major with K % 8 == 0; outputs fp32 (C) and/or bf16 (Cb); same epilogue
 * as dmm_gemm.  dmm_cast_bf16 makes the bf16 copies (weights once per epoch, x_0 once per chain); the time embedding of a chain
 * step (all rows share t) enters as that step's bias row of dmm_temb_bias_table.  Tile choice: K % 64 == 0 and >= 64 tiles of
 * 256 x 256 -> the LDS-DMA tile of csrc/gemm_bf16_8ph.hip; otherwise the register-staged 128 x 64 / 128 x 128 / 256 x 128 tiles. */
typedef struct dmm_gemm_bf16_args {
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  float* C; int64_t ldc;
  void* Cb; int64_t ldcb;
  const float* bias;
  const float* R; int64_t ldr;
  const void* Rb; int64_t ldrb; /* residual in bf16 (used when R is NULL) */
  int32_t M, N, K;
  float alpha, e1, e2;
  int32_t act;
  const float* bias2; /* [N] or NULL.  Non-NULL (needs C and Cb): Cb = tanh(C + bias2[n]) instead of a bf16 copy of C -- a step of the
                       * latent reverse chain writes its fp32 state z' and the next step's layer-1 activation at once */
} dmm_gemm_bf16_args;
int dmm_gemm_bf16_nt(const dmm_gemm_bf16_args* a /*host*/, void* stream);
/* up to 4 problems of one shape in one launch (the modalities' large GEMMs go out together: enough tiles to fill the chip) */
int dmm_gemm_bf16_nt_batched(const dmm_gemm_bf16_args* a /*host array*/, int n_batch, void* stream);

/* The latent reverse chain as one persistent launch sequence: GaussianDiffusion.p_sample's T posterior-mean steps (Model.py:300-322,
 * :375) for up to 4 modalities.  With z_t = x_t W1x^T, G = W1x W2, g = W1x b2:   z_{t-1} = c2[t] z_t + c1[t] (h_t G^T + g),  h_{t-1} = tanh(z_{t-1} +
 * bias_t[t-1]).  On entry z holds z_T (fp32) and h_a holds h_{T-1} (bf16); the T - 1 steps t = T-1 .. 1 run inside the kernel with z in registers
 * (it is never written back) and h exchanged through h_a / h_b; on return h_0 is in h_a when T - 1 is even, else in h_b.  The caller then forms
 * x_0' = c1[0] (h_0 W2^T + b2) with dmm_gemm_bf16_nt.  coef (device, [2 T]): coef[2 t] = c1[t], coef[2 t + 1] = c2[t-1] c1[t] / c1[t-1];
 * s_init = c2[T-1] / c1[T-1].  flags: device int32 scratch, >= 514 entries; flags[flags_len - 1] is an error word: bit 0 = a group barrier timed
 * out (the results are then invalid; the kernel never hangs), bit 1 = the workgroups of a group did not share an XCD (results valid, the kernel's
 * L2 re-read assumption did not hold).  H must be a multiple of 256 (<= 2048). */
typedef struct dmm_latent_chain_args {
  const void* G; int64_t ldg;      /* bf16 [H, H] */
  const float* g;                  /* [H] */
  const float* z; int64_t ldz;     /* fp32 [rows, H] */
  void* h_a; void* h_b; int64_t ldh; /* bf16 [rows, H] each */
  const float* bias_t; int64_t ldbt; /* fp32 [T, H]: per-timestep layer-1 bias (dmm_temb_bias_table) */
} dmm_latent_chain_args;
int dmm_latent_chain(const dmm_latent_chain_args* a /*host array*/, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                     int flags_len, void* stream);
/* 1 when the current device can keep a launch slice of the persistent kernel co-resident (>= 8 * H / 256 CUs, one workgroup of it fits a CU), which
 * its group barrier needs; 0 otherwise -- dmm_latent_chain then returns DMM_ERR_UNSUPPORTED without launching and the caller runs one
 * dmm_gemm_bf16_nt launch per step.  (Host query, no stream; DMM_LATENT_CUS=<n> overrides the CU count on a masked / partitioned device.) */
int dmm_latent_chain_supported(int H);
/* The chain one LAUNCH at a time.  dmm_latent_chain runs ceil(n_mod * ceil(rows / 256) / G) launches of G = dmm_latent_chain_groups_per_launch(H)
 * row-block groups each (modality-major: launch p holds groups p G .. (p + 1) G - 1; 0 = the persistent form cannot run or slices by itself);
 * dmm_latent_chain_part issues launch `part` alone, so that the host can put the finished modalities' last GEMM and top-k on another stream UNDER
 * the next launch, which leaves CUs idle (111 groups at the TikTok shape = 64 + 47).  Parts must be issued in order on one stream. */
int dmm_latent_chain_groups_per_launch(int H);
int dmm_latent_chain_part(const dmm_latent_chain_args* a /*host array*/, int n_mod, int rows, int H, int T, const float* coef, float s_init, int32_t* flags,
                          int flags_len, int part, void* stream);
int dmm_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int cols_pad, void* stream);
/* dst[c, r] = bf16(src[r, c]) for r < rows; dst[c, rows..rows_pad) = 0 (makes the K-contiguous operand of a TN / NN product) */
int dmm_transpose_cast_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int rows_pad, void* stream);
/* The same casts keeping what the rounding lost: lo = bf16(x - float(hi)), laid out like hi (hi + lo carries 16 mantissa bits).  Used once per
 * epoch for the latent chain's G = W1x W2: hi*hi + lo*hi + hi*lo on the bf16 MFMA GEMM is fp32-grade before G's own rounding. */
int dmm_cast_bf16_split(const float* src, int64_t lds, void* hi, void* lo, int64_t ldd, int rows, int cols, int cols_pad, void* stream);
int dmm_transpose_cast_bf16_split(const float* src, int64_t lds, void* hi, void* lo, int64_t ldd, int rows, int cols, int rows_pad, void* stream);
/* out[r] = sum_c W[r, c] x[c], fp32 (the latent chain's g = W1x b2) */
int dmm_gemv_rows(const float* W, int64_t ldw, const float* x, float* out, int rows, int cols, void* stream);
/* table[t][h] = b1[h] + sum_c W1[h][I+c] * emb_layer(features(t))[c]: the layer-1 bias of chain step t (all rows share t), T x H */
int dmm_temb_bias_table(const float* w_emb, const float* b_emb, int te, const float* w1, int64_t ldw1, int I, const float* b1, int H,
                        int T, float* table, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Row L2 normalisation, F.normalize(x, p=2, dim=1, eps=1e-12): Model.py:89,92,104; Utils.py:72.
 * fwd writes y = x / max(||x||, eps) and norm = ||x||.  bwd: dx = (dy - y (y.dy)) / ||x||  (dy / eps below eps). */
int dmm_rownorm_fwd(const float* x, float* y, float* norm, int rows, int D, void* stream);
int dmm_rownorm_fwd_x16(const float* x, float* y, void* y16 /*bf16 copy of y*/, float* norm, int rows, int D, void* stream);
int dmm_rownorm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D, void* stream);
/* dmm_rownorm_bwd for up to 4 tensors in one launch; csum (optional, [D], pre-zeroed) receives the column sums of dx -- the bias gradient of the
 * Linear layer in front of the normalisation (Model.py:84-89): the joint step's three rownorm_bwd + three colsum launches in one */
typedef struct dmm_rownorm_bwd_problem {
  const float* dy; const float* y; const float* norm;
  float* dx; float* csum;
  int32_t rows;
} dmm_rownorm_bwd_problem;
int dmm_rownorm_bwd_batched(const dmm_rownorm_bwd_problem* p /*host array*/, int n, int D, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Modal mixing of gcn_MM, Model.py:87,116-127:  mx = sum_m softmax(w)_m * (bx + maw * z_m).
 * bwd: dbx = (sum_m w_m) dmx + dbx_extra;  dz_m += maw * w_m * dmx;  dlogits = softmax-bwd(<dmx, bx + maw z_m>).
 * `partials` is scratch of dmm_gcn_mix_partials(n) floats. */
int dmm_gcn_mix_fwd(const float* bx, const float* const* z /*host array of M device ptrs*/, const float* w_logits, int M,
                    float maw, float* mx, int64_t n, void* stream);
int64_t dmm_gcn_mix_partials(int64_t n);
int dmm_gcn_mix_bwd(const float* dmx, const float* bx, const float* const* z /*host*/, float* const* dz /*host, in/out*/,
                    const float* dbx_extra, float* dbx, const float* w_logits, float* dw_logits, int M, float maw,
                    float* partials, int64_t n, void* stream);
/* The same, but only elements [0, count0_end) and [count1_begin, count1_end) enter d(logits): a user-sharded replica counts its own
 * user rows and its slice of the replicated item rows, and the replicas' dw_logits add up (the softmax backward is linear in the
 * per-modality dot products) to the unsharded gradient. */
int dmm_gcn_mix_bwd_counted(const float* dmx, const float* bx, const float* const* z /*host*/, float* const* dz /*host, in/out*/,
                            const float* dbx_extra, float* dbx, const float* w_logits, float* dw_logits, int M, float maw,
                            float* partials, int64_t n, int64_t count0_end, int64_t count1_begin, int64_t count1_end, void* stream);

/* bf16 gather operands of the joint phase's SpMMs (graphs planned with vec = 8, dmm_spmm_csr_bf16) made where their fp32 source is produced:
 * the mixing kernels also write bf16 copies of their outputs (mx16; dbx16 and the updated dz16[m]) ... */
int dmm_gcn_mix_fwd_x16(const float* bx, const float* const* z /*host*/, const float* w_logits, int M, float maw, float* mx, void* mx16 /*bf16 [n]*/,
                        int64_t n, void* stream);
int dmm_gcn_mix_bwd_x16(const float* dmx, const float* bx, const float* const* z /*host*/, float* const* dz /*host, in/out*/, const float* dbx_extra,
                        float* dbx, const float* w_logits, float* dw_logits, int M, float maw, float* partials, int64_t n, int64_t count0_end,
                        int64_t count1_begin, int64_t count1_end /*as dmm_gcn_mix_bwd_counted; n, n, n = everything*/, void* dbx16,
                        void* const* dz16 /*host array of M bf16 buffers*/, int dz_overwrite /*1: dz_m = maw w_m dmx, dz is not read*/, void* stream);
/* ... and the gradient tensors the losses scatter into (zero outside the minibatch's rows users[b], U + pos[b], U + neg[b]: Model.py:84-134's
 * outputs indexed by ancs / poss / negs, Main.py:318-352) get their bf16 copies by casting just those rows of up to 4 [N, D] tensors; src == NULL or
 * src[t] == NULL writes zeros (clearing the rows again after use keeps the copy zero elsewhere).  D % 8 == 0, 256 % (D / 8) == 0. */
/* the same rows of up to 8 tensors cleared, row_bytes[t] bytes per row (multiples of 16): fp32 and bf16 tensors in one launch */
/* seg (device, int32[2], or NULL = all): a user-sharded replica's users[] holds LOCAL row ids that are valid only for b in [seg[0], seg[1]) -- the
 * minibatch entries whose user it owns; the item rows U + pos[b], U + neg[b] (U = its own user count) are touched for every b. */
int dmm_zero_batch_rows(void* const* dst /*host*/, const int32_t* row_bytes /*host*/, int n_t, const int32_t* users, const int32_t* pos, const int32_t* neg,
                        int B, int U, const int32_t* seg, void* stream);
int dmm_cast_batch_rows_bf16(const float* const* src /*host array or NULL*/, void* const* dst /*host array of n_t bf16 [N, D]*/, int n_t, int D,
                             const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int U, const int32_t* seg, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Contrastive + ranking losses.  Utils/Utils.py:57-75 (InfoNCE), :78-98 (bpr_loss), :45-54 (l2_reg_loss).
 * Views are gathered and row-normalised once per (tensor, index) pair ("slot"); a term is a pair of slots.
 *   loss_t = weight_t * mean_i( lse_i - s_ii ),  s = zhat_a zhat_b^T / temp
 * dmm_infonce_bwd accumulates d(zhat) for every slot (dzhat must be zeroed by the caller);
 * dmm_scatter_normalize_bwd pushes it through the normalisation and adds it to grad[idx[b]] (fp32 atomics,
 * duplicate indices allowed -- the backward of view[idx]). */
typedef struct dmm_gather_desc {
  const float* src;   /* [rows, D] */
  const int32_t* idx; /* [B] */
  float* grad;        /* [rows, D] gradient buffer of src (used by dmm_scatter_normalize_bwd) or NULL */
} dmm_gather_desc;
typedef struct dmm_infonce_term {
  int32_t a, b; /* slot ids */
  float temp, weight;
} dmm_infonce_term;
/* bf16_pack (optional, dmm_infonce_pack_bytes() bytes, 16-byte aligned): bf16 hi/lo copies of zhat, row-major and transposed,
 * padded to 32 rows -- the operands of the MFMA InfoNCE kernels.  precision: 0 = fp32 VALU kernels (any D % 4 == 0),
 * 1 = one bf16 MFMA product, 3 = hi/lo split (~fp32); the MFMA kernels need D in {32, 64, 128, 256} and the pack. */
int64_t dmm_infonce_pack_bytes(int n_slots, int B, int D);
int dmm_gather_normalize(const dmm_gather_desc* slots /*host*/, int n_slots, int B, int D, float* zhat /*[n_slots,B,D]*/,
                         float* norm /*[n_slots,B]*/, void* bf16_pack, void* stream);
/* The same with only the hi / hiT blocks of the pack written: for callers whose InfoNCE launches run at precision 1 (the lo / loT blocks feed the
 * split-precision kernels only; their contents are then undefined). */
int dmm_gather_normalize_hi(const dmm_gather_desc* slots /*host*/, int n_slots, int B, int D, float* zhat, float* norm, void* bf16_pack, void* stream);
int dmm_infonce_fwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const dmm_infonce_term* terms /*host*/,
                    int n_terms, int B, int D, float* lse /*[n_terms,B]*/, float* row_loss /*[n_terms,B], weight/B folded in*/,
                    void* stream);
int dmm_infonce_bwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const float* lse,
                    const dmm_infonce_term* terms /*host*/, int n_terms, int B, int D, float* dzhat /*[n_slots,B,D], accumulated*/,
                    void* stream);
/* forward and backward of all terms in one call, for UNIT-NORM rows (dmm_gather_normalize's output; cosine InfoNCE, Utils.py:60-62): with every
 * logit <= 1 the exponentials need no running maximum, so ONE walk over the other view gives lse, row_loss and the anchor-side gradient, and the
 * backward keeps only the pass for the other view (two kernels instead of forward + a two-pass backward that recomputed every logit twice).
 * dzhat must be zeroed by the caller.  The one-walk kernels run when DMM_INFONCE_FUSED=1 (measured slower at B = 4096: too few waves per launch;
 * csrc/losses.hip); otherwise, and for fp32-VALU arithmetic or temperatures < 0.024, this is dmm_infonce_fwd followed by dmm_infonce_bwd. */
int dmm_infonce_fwd_bwd(const float* zhat, const void* bf16_pack, int n_slots, int precision, const dmm_infonce_term* terms /*host*/, int n_terms,
                        int B, int D, float* lse, float* row_loss, float* dzhat, void* stream);
int dmm_scatter_normalize_bwd(const dmm_gather_desc* slots /*host*/, int n_slots, int B, int D, const float* zhat,
                              const float* norm, const float* dzhat, void* stream);
/* BPR forward + backward in one pass: row_loss[b] = -log(1e-5 + sigmoid(u.p - u.n)) / B and, when dfin != NULL,
 * dfin[user/pos/neg rows] += d(row_loss) (atomics).  `fin` is [U + I, D]; item i lives at row U + i. */
int dmm_bpr_fwd_bwd(const float* fin, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int D, int U,
                    float* row_loss, float* dfin, void* stream);
/* The same with the user rows and the item rows in two tables (users[b] indexes fin_u / dfin_u, pos[b] and neg[b] index fin_i / dfin_i):
 * the user-sharded engine scores the exchanged batch table of user rows against its replicated item rows. */
int dmm_bpr_fwd_bwd_tables(const float* fin_u, const float* fin_i, const int32_t* users, const int32_t* pos, const int32_t* neg, int B, int D,
                           float* row_loss, float* dfin_u, float* dfin_i, void* stream);
/* User-sharded batch exchange (replaces nothing in the reference, which is single-process: Main.py:305-307 gathers whole
 * tables).  A replica owns the user rows of the batch positions [seg[0], seg[1]) (seg: DEVICE int32[2], read by the kernel so the launch
 * is the same every step; NULL = all of [0, B)).
 *   dmm_gather_rows       out[s][b] = slots[s].src[slots[s].idx[b]]   for b in the segment; other rows of out [n_slots, B, D] untouched
 *                         (zero them first: a sum all-reduce over the replicas then assembles every replica's copy of the table)
 *   dmm_scatter_add_rows  slots[s].grad[slots[s].idx[b]] += in[s][b]  for b in the segment (fp32 atomics; repeated users are legal; a slot whose
 *                         grad is NULL is skipped) */
int dmm_gather_rows(const dmm_gather_desc* slots /*host*/, int n_slots, int B, int D, const int32_t* seg, float* out, void* stream);
int dmm_scatter_add_rows(const dmm_gather_desc* slots /*host*/, int n_slots, int B, int D, const int32_t* seg, const float* in, void* stream);
/* out[s] = scale * sum(in[s*len .. (s+1)*len)) in a fixed order (deterministic loss reduction). */
int dmm_segment_sum(const float* in, int n_seg, int64_t len, float scale, float* out, void* stream);
/* out[0] = scale * sum(x^2) (l2_reg_loss, Utils.py:45-54); scratch: dmm_sumsq_scratch(n) floats. */
int64_t dmm_sumsq_scratch(int64_t n);
int dmm_sumsq(const float* x, int64_t n, float scale, float* scratch, float* out, void* stream);
/* dmm_segment_sum(seg_in ...) and dmm_sumsq(x ...) of the joint step (the InfoNCE / BPR row losses and the l2 term, Main.py:353-362) in ONE launch:
 * seg_out[k] = seg_scale * sum(seg_in[k * len .. ]), x_out[0] = x_scale * sum(x^2) -- the same sums in the same order.  scratch: dmm_sumsq_scratch(n) + 1
 * floats, the last one zero before the first call (an arrival counter the kernel resets).  acc (optional, device fp64): running sums over the steps
 * of an epoch, acc[k] += seg_out[k], acc[acc_x] += x_out[0]. */
int dmm_loss_reduce(const float* seg_in, int n_seg, int64_t len, float seg_scale, float* seg_out, const float* x, int64_t n, float x_scale, float* scratch,
                    float* x_out, double* acc, int acc_x, void* stream);
/* Up to 4 regions (16-byte aligned, whole 16-byte pieces) cleared by one launch. */
int dmm_zero_regions(void* const* ptrs /*host*/, const int64_t* bytes /*host*/, int n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults: betas 0.9/0.999, eps 1e-8, weight_decay 0; Main.py:92-110,189-192,377).
 * g_total = g + sum_k extra_k (first extra_n[k] elements only) + l2_coef * p -- lets the embedding update absorb
 * the user-row parts of the modal SpMM backward and the 2*reg*p term of l2_reg_loss without extra passes.
 * bias corrections use `step` (1-based). */
int dmm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra /*host, <=4*/,
                  const int64_t* extra_n /*host*/, int n_extra, float l2_coef, float lr, float beta1, float beta2,
                  float eps, int step, void* stream);
/* the same step with the l2 term applied to the first l2_n elements only (one launch for a flat buffer = [embedding rows | other weights]) */
int dmm_adam_step_prefix(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra /*host, <=4*/, const int64_t* extra_n,
                         int n_extra, float l2_coef, float lr, float beta1, float beta2, float eps, int step, void* stream, int64_t l2_n);
/* ... and with a bf16 copy of the updated parameters p[p16_lo .. p16_lo + p16_n) written to p16[0 .. p16_n) (the embedding table's gather copy for
 * dmm_spmm_csr_bf16; a Denoise's W1 | W2 as the next step's bf16 GEMM operands: made by the step that changes them instead of a cast pass per step) */
int dmm_adam_step_x16(float* p, const float* g, float* m, float* v, int64_t n, const float* const* extra /*host, <=4*/, const int64_t* extra_n,
                      int n_extra, float l2_coef, float lr, float beta1, float beta2, float eps, int step, void* stream, int64_t l2_n, void* p16,
                      int64_t p16_n, int64_t p16_lo);

/* ---------------------------------------------------------------------------------------------------
 * Diffusion path.
 * dmm_qsample: x_t = a[t_b] x0 + b[t_b] eps  (GaussianDiffusion.forward_cal_xt, Model.py:324-341) written into the
 * first I columns of xcat [B, ldx]; columns I..I+te get the time embedding emb_layer(cat[cos, sin](t f_k))
 * (Denoise.forward, Model.py:196-202), the rest of the row up to ldx is zeroed.  eps read from `noise` [B, I]
 * or drawn in-kernel (Philox + Box-Muller) when noise == NULL and draw_noise != 0; with both off, eps = 0 and
 * a = 1 (the reverse chain's x_t passes through).  `eps_out` (optional) receives the noise used. */
int dmm_qsample_temb(const float* x0, int64_t ldx0, const float* noise, const int32_t* t, const float* sqrt_ac,
                     const float* sqrt_1mac, const float* w_emb /*[te,te]*/, const float* b_emb /*[te]*/, int te,
                     float* xcat, int64_t ldx, int B, int I, int draw_noise, uint64_t seed, uint64_t offset,
                     float* eps_out, void* stream);
/* x_0 as CSR rows.  The reference keeps the dense U x I interaction matrix on the device (DataHandler.py:128) and indexes rows out of it
 * (DataHandler.py:218-224); here the interactions stay CSR (ptr int64 [U+1] over ALL users, col int32) and `rows` (int64 [B]) names the users:
 *   dmm_csr_dense_rows       out[b, :ldo] = the 0/1 row of user rows[b] (columns >= I zero) -- the dense operand where one is needed;
 *   dmm_csr_gather_sum_bf16  out16[b, :H] = act(bias + sum_{j in items(rows[b])} Wt[j, :H]), Wt = W^T in bf16 [>= I, ldw]: a Linear layer applied
 *                            to a 0/1 row is a gather of W^T rows (Denoise layer 1 of the first reverse-chain step, x_T = x_0: Model.py:300-322, :211);
 *   dmm_csr_add_rows         y[b, j] += c (and y16[b, j] = bf16 of it) for j in items(rows[b]): the posterior-mean residual c2 * x_0 of that step
 *                            (Model.py:375) without a dense x_0; y == NULL: the row is kept in bf16 only (y16 read and written). */
int dmm_csr_dense_rows(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, float* out, int64_t ldo, void* stream);
/* the same for a slice of the item columns: items col_lo .. col_lo + I - 1 become columns 0 .. I - 1 (item-sharded Denoise: a replica holds the
 * W1 columns / W2 rows of its item slice and reads x_0 restricted to it) */
int dmm_csr_dense_rows_slice(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int col_lo, int I, float* out, int64_t ldo, void* stream);
/* x_0 rows as BIT MASKS: bits[b, w] bit k = 1 iff item 32 w + k is an interaction of user rows[b] (ldb words per row, ldb * 32 >= I; the words'
 * unused bits are zero).  The training step's three readers of the dense 0/1 rows take the masks instead -- 1/32 of the bytes:
 * dmm_qsample_temb_bits, dmm_diff_loss_bits (forward loss), dmm_item_expand_fused mode 1 (X0bits). */
int dmm_csr_row_bits(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, uint32_t* bits, int64_t ldb, void* stream);
int dmm_csr_row_bits_slice(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int col_lo, int I, uint32_t* bits, int64_t ldb, void* stream);
int dmm_qsample_temb_bits(const uint32_t* x0bits, int64_t ldb, const float* noise, const int32_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                          const float* w_emb, const float* b_emb, int te, float* xcat, int64_t ldx, int B, int I, int draw_noise, uint64_t seed,
                          uint64_t offset, float* eps_out, void* stream);
int dmm_diff_loss_bits(const float* out, int64_t ld, const uint32_t* x0bits, int64_t ldb, const double* w, const float* um, const float* ui, int B, int I,
                       int d, float sim_weight, double reg_term, double* loss /*[B]*/, void* stream);
int dmm_csr_gather_sum_bf16(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const void* Wt, int64_t ldw, const float* bias, int H,
                            int act, void* out16, int64_t ldo, void* stream);
int dmm_csr_add_rows(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, int I, float c, float* y, int64_t ldy, void* y16, int64_t ldy16,
                     void* stream);
/* start of the latent reverse chain (GaussianDiffusion.p_sample with sampling_steps == 0, Model.py:300-322): z[b, :H] = sum_{j in
 * items(rows[b])} Wt[j, :H] (fp32, = x_0 W1x^T) and out16 = bf16(tanh(z + bias)), the first step's layer-1 activation */
int dmm_csr_gather_sum_latent(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const void* Wt, int64_t ldw, const float* bias, int H,
                              float* z, int64_t ldz, void* out16, int64_t ldo, void* stream);
/* The same start for up to 4 operand tables in ONE launch (the M modalities' chains start from the same rows). */
typedef struct dmm_csr_gather_problem {
  const void* Wt; int64_t ldw;   /* bf16 [I, >= H]: W1x^T */
  const float* bias;             /* [H]: bias_t[T - 1] */
  float* z; int64_t ldz;         /* fp32 [B, >= H] out: z_T */
  void* out16; int64_t ldo;      /* bf16 [B, >= H] out: h_{T-1} = tanh(z_T + bias) */
} dmm_csr_gather_problem;
int dmm_csr_gather_sum_latent_batched(const int64_t* ptr, const int32_t* col, const int64_t* rows, int B, const dmm_csr_gather_problem* p /*host array*/, int n,
                                      int H, void* stream);
/* out[b, :I] = a_table[t_b] * x[b, :I] + c_table[t_b] * y[b, :I] (per-row timesteps): forward_cal_xt with explicit noise (Model.py:339-341)
 * and the posterior mean c1[t] * Denoise(x_t, t) + c2[t] * x_t of p_mean_variance (Model.py:375), for the reference-named Python API
 * (the training / rebuild engines fuse both into their GEMM operands and epilogues). */
int dmm_rows_axpby(const float* x, int64_t ldx, const float* y, int64_t ldy, const int32_t* t, const float* a_table, const float* c_table,
                   float* out, int64_t ldo, int B, int I, void* stream);
/* Backward of the time-embedding linear layer: d w_emb += temb_in^T d(temb), d b_emb += colsum (one block). */
int dmm_temb_bwd(const int32_t* t, const float* dxcat, int64_t ldx, int I, int te, int B, float* dw_emb, float* db_emb,
                 void* stream);
/* Elementwise pieces of Denoise / training_losses (Model.py:204-218, 407-425). */
int dmm_mul_sigmoid_fwd(const float* proj, const float* gate_pre_sig /*sigmoid already applied*/, float* out, int64_t n,
                        void* stream); /* out = proj * gate */
int dmm_gate_bwd(const float* dout, const float* proj, const float* gate, float* dproj /*out: dout*gate*/,
                 float* dgate_pre /*out: dout*proj*gate*(1-gate)*/, int64_t n, void* stream);
/* TrainData.negSampling (DataHandler.py:159-169): out[i] = an item drawn uniformly from those user users[i] has no train interaction with -- the
 * distribution of the reference's rejection loop, sampled without rejection (the r-th non-positive item by binary search over the user's sorted
 * positives ptr / col_sorted, CSR over all users) from Philox4x32-10(seed, offset + i): one launch, no host round trip, no generator state. */
int dmm_neg_sample(const int64_t* ptr, const int32_t* col_sorted, const int32_t* users, int n, int items, uint64_t seed, uint64_t offset, int32_t* out,
                   void* stream);
int dmm_tanh_bwd(const float* dh, const float* h, float* dpre, int64_t n, void* stream); /* dpre = dh * (1 - h^2) */
/* The same product as the operands the layer-1 weight-gradient GEMMs want, in ONE pass over dh and h for up to 4 problems (the modalities of a
 * diffusion-training step): dpre = dh * (1 - h^2) is written only as bf16 rows d16 [rows, cols], as its bf16 transpose dT16 [cols, rows_pad]
 * (columns rows..rows_pad zero: the K-contiguous operand of an NT GEMM with K = rows_pad, a multiple of 8) and added, column-summed, to csum [cols]
 * (fp32 atomics: the bias gradient).  Replaces dmm_tanh_bwd + dmm_transpose_cast_bf16 + dmm_colsum_add + dmm_cast_bf16 (4 launches and an fp32
 * round trip of dpre per problem); the bf16 values are bit-identical to theirs. */
typedef struct dmm_tanh_bwd_fused_args {
  const float* dh; int64_t ld_dh;
  const float* h; int64_t ld_h;
  void* d16; int64_t ld16;    /* bf16 [rows, cols] */
  void* dT16; int64_t ldT;    /* bf16 [cols, >= rows_pad] */
  float* csum;                /* [cols], accumulated into */
} dmm_tanh_bwd_fused_args;
int dmm_tanh_bwd_fused(const dmm_tanh_bwd_fused_args* a /*host*/, int n, int rows, int cols, int rows_pad, void* stream);
/* x = tanh(x + bias[col]) in place on a contiguous fp32 [rows, cols] (cols % 4 == 0), out16 (optional): its bf16 copy.  Denoise layer 1's activation
 * (Model.py:212) applied AFTER the item-sliced partial products x_t[:, I_g] W1[:, I_g]^T have been summed over the replicas. */
int dmm_bias_tanh_rows(float* x, const float* bias, int rows, int cols, void* out16 /*bf16 [rows, cols] or NULL*/, void* stream);
int dmm_add_inplace(float* y, const float* x, float alpha, int64_t n, void* stream);   /* y += alpha * x */
int dmm_fill_zero(void* p, int64_t bytes, void* stream);                                /* hipMemsetAsync(p, 0, bytes) */
int dmm_colsum(const float* x, int64_t ldx, int rows, int cols, float* out /*[cols], overwritten*/, void* stream);
/* the same sums ADDED to out (no memset launch: for outputs inside a region the caller has just zeroed, e.g. the bias gradients of nn.Linear) */
int dmm_colsum_add(const float* x, int64_t ldx, int rows, int cols, float* out /*[cols], accumulated into*/, void* stream);
/* Diffusion loss, Model.py:407-425.  per row b: mse_b = mean_j (out - x0)^2; cos_b = cosine(um_b, ui_b);
 * loss_b = w_b * mse_b + sim_weight * (1 - cos_b) + reg_term  (float64 out, like the reference).
 * Also emits d loss / d out (the direct MSE part) and d loss / d um scaled by gscale (= 1 / (B * total)).
 * loss == NULL and dout == NULL: only d loss / d um (the [B, I] rows are not read; dmm_item_expand_fused mode 1 forms d loss / d out). */
int dmm_diff_loss(const float* out, const float* x0, int64_t ld, const double* w /*[B]*/, const float* um,
                  const float* ui, int B, int I, int d, float sim_weight, double reg_term, double gscale,
                  double* loss /*[B] or NULL*/, float* dout /*[B,ld] or NULL*/, float* dum /*[B,d] or NULL*/,
                  const double* gscale_dev /*NULL, or a device scalar multiplied into gscale: Main.py:176 divides every modality's loss by the minibatch's
                                             loss total before backward -- with it the step needs no host read of that total*/, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Rebuild: per-row top-k with k = the user's degree (Main.py:224-230).  One workgroup per row; writes the k
 * selected column ids of row r to out_idx[out_ptr[r] .. out_ptr[r] + k) (ascending column order; ties at the
 * threshold resolved towards the smaller column id). */
int dmm_topk_rows(const float* x, int64_t ldx, const int32_t* row_ids /*[rows] global user ids*/, const int32_t* out_ptr
                  /*[U+1] indexed by user id*/, int32_t* out_idx, int rows, int cols, void* stream);

/* Start of a reverse chain with sampling_step > 0 (generate_view -> forward_cal_xt without noise, Model.py:310-313,336-341):
 * xT[b, :I] = sqrt_ac * x0 + sqrt_1mac * sign(x0) * n / max(||n||_2, 1e-12), n ~ N(0, 1)^I per row (in-kernel Philox) or `randn`
 * ([B, I] contiguous) when given. */
int dmm_noised_start(const float* x0, int64_t ldx0, const float* randn, float sqrt_ac, float sqrt_1mac, float* xT, int64_t ldx, int B, int I,
                     uint64_t seed, uint64_t offset, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Item-space products of the diffusion training step (bf16-rounded operands, fp32 accumulation; D = 64, 128 or 256):
 *   dmm_item_project  out[B, D] += X[B, :I] . F[I, D]   -- Model.py:207 (x_t @ modal_feat), :413 (model_output @ model_feats), :414
 *                     (x_start @ itmEmbeds).  `out` is accumulated with atomics: zero it first.  X rows must be readable (finite
 *                     values) up to round_up(I, 32) columns; those columns are multiplied by zero.
 *   dmm_item_expand   Y[B, :I] = R[B, :I] + P[B, D] . F[I, D]^T  -- Model.py:209 (x_t + (proj * gate) @ modal_feat.T); Y may alias R. */
int dmm_item_project(const float* X, int64_t ldx, const float* F, int64_t ldf, float* out, int64_t ldo, int B, int I, int D, void* stream);
int dmm_item_expand(const float* P, int64_t ldp, const float* F, int64_t ldf, const float* R, int64_t ldr, float* Y, int64_t ldy, int B, int I, int D,
                    void* stream);
/* dmm_item_expand fused with its consumers in the diffusion training step: the [B, I] result is written only as the bf16 operands of the next
 * GEMMs -- row-major Y16[B, Kc] and transposed YT[Kt, round_up(B, 64)] (batch rows >= B zero) -- never in fp32.
 *   mode 0 (Model.py:209):      v = R[b, c] + (P F^T)[b, c] for c < I;  v = R[b, c] for I <= c < Kc  (the [x_t | temb | 0] operand of layer 1)
 *   mode 1 (Model.py:418-428):  v = gscale w[b] 2/I (R[b, c] - X0[b, c]) + (P F^T)[b, c] for c < I, 0 beyond: the gradient of the diffusion loss
 *                               w.r.t. the Denoise output (R = that output, P = d loss / d (out F)), plus csum[c] += sum_b v (bias gradient) */
typedef struct dmm_expand_fused_args {
  const float* P; int64_t ldp;
  const float* F; int64_t ldf;
  const float* R; int64_t ldr;
  const float* X0; int64_t ldx0; /* mode 1 */
  const double* w; double gscale; /* mode 1 */
  void* Y16; int64_t ldy;        /* bf16 [B, >= Kc] */
  void* YT; int64_t ldt;         /* bf16 [Kt, >= round_up(B, 64)] */
  float* csum;                   /* mode 1: [I], accumulated into */
  int32_t B, I, Kc, Kt;
  const uint32_t* X0bits; int64_t ldbits; /* mode 1, X0 == NULL: x_0 as bit masks (dmm_csr_row_bits), ldbits words per row */
  const double* gscale_dev;               /* mode 1: NULL, or a device scalar multiplied into gscale (see dmm_diff_loss) */
  const void* F16; int64_t ldf16;         /* optional: bf16 copy of F [I, >= D] (dmm_cast_bf16 of it): staged instead of F -- same values, half the bytes */
} dmm_expand_fused_args;
int dmm_item_expand_fused(const dmm_expand_fused_args* a /*host*/, int mode, int D, void* stream);
/* The same product for up to 4 problems in one launch, 64 rows per workgroup (many short rows), with the layouts of the joint step's
 * modality-feature projections (Model.py:48-57,84-85,103):   out[r, c] += sum_k X[r, k] Fm(k, c) (+ bias[c], added once)
 *   f_transposed:   Fm(k, c) = F[c * ldf + k]  (F is an nn.Linear weight [D, K]);  else F[k * ldf + c]
 *   out_transposed: the result is stored at out[c * ldo + r]  (the weight gradient dW[D, F] = dfeat^T E  as  E^T[F, I] x dfeat[I, D])
 * out must hold the value to accumulate into (K is cut into slices added with fp32 atomics); X rows readable up to round_up(K, 32). */
typedef struct dmm_project_problem {
  const float* X; int64_t ldx;
  const float* F; int64_t ldf;
  float* out; int64_t ldo;
  const float* bias; /* [D] or NULL */
  int32_t rows, K;
} dmm_project_problem;
int dmm_item_project_batched(const dmm_project_problem* p /*host array*/, int n, int D, int f_transposed, int out_transposed, void* stream);
/* the out_transposed form with X given in bf16 ([rows, ldx] bf16 elements, ldx % 8 == 0; the X pointer of the problem struct is read as such):
 * a transposed bf16 weight copy that exists anyway as the long operand, e.g. (F^T W2)[d, H] from W2^T[H, I] in bf16 (Model.py:413 folded) */
int dmm_item_project_batched_x16(const dmm_project_problem* p /*host array*/, int n, int D, void* stream);
/* The same product with X given in bf16 (a constant operand cast once: the raw modality features of Model.py:34-42's Linear layers and their
 * transpose in the joint step) -- bit-identical to the fp32-X forms at precision 1, which round X to bf16 inside the kernel, at half the bytes.
 * Forms: (f_transposed = 1, out_transposed = 0) and (0, 1). */
int dmm_item_project_batched_xb(const dmm_project_problem* p /*host array*/, int n, int D, int f_transposed, int out_transposed, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Rebuild, second half: the modality graph of Main.py:232-253 (coo_matrix of the top-k edges -> DataHandler.makeTorchAdj,
 * DataHandler.py:54-93) built on the device, together with the work list dmm_spmm_csr_f32 consumes.
 *   user_ptr  int32 [U+1]  CSR of the user rows (k_u = train degree, fixed for the run);  edge_user int32 [E] = user of edge e
 *   item_idx  int32 [E]    user-major item ids as dmm_topk_rows wrote them (distinct and ascending inside a user)
 *   dinv_table f64 [>= max row length + 1]  entry d = d^-0.5 (host-computed, so values round exactly like the scipy path)
 * Outputs (caller-allocated): rowptr int32 [U+I+1], col int32 [nnz], val f32 [nnz] with nnz = 2E + U + I (self loops included,
 * columns ascending in every row), tasks int4 [dmm_graph_build_task_capacity], fix int4 [...], meta int32 [8] =
 * {n_tasks, n_head_tasks, n_partial_blocks, n_fix, nnz, ...}.  Bit-identical to diffmm_amd/graph.py's host builder + planner. */
int dmm_graph_build_scratch_bytes(int U, int I, int64_t E, int D, int64_t* bytes);
int dmm_graph_build_task_capacity(int U, int I, int64_t E, int D, int task_len, int64_t* tasks, int64_t* fix);
int dmm_graph_build(const int32_t* user_ptr, const int32_t* edge_user, const int32_t* item_idx, int U, int I, int64_t E, int D, int task_len,
                    const double* dinv_table, int32_t* rowptr, int32_t* col, float* val, void* tasks, void* fix, int32_t* meta, void* scratch,
                    void* stream);
/* User-sharded variant: builds a replica's local block over the node order [own users (U of them) ; all items].  user_ptr /
 * edge_user / item_idx describe the replica's OWN users only; item_count_global int32 [I] = number of users, over ALL replicas, that
 * selected each item (dmm_count_items on every replica, then a sum all-reduce) -- it sets the D^-1/2 factors, so the block carries the
 * values of the full graph; item i's self loop is emitted only where i % self_mod == self_rank (so the replicas' blocks add up to the
 * full matrix, each entry exactly once).  nnz <= 2E + U + I.  item_count_global == NULL, self_mod = 1: identical to dmm_graph_build. */
int dmm_count_items(const int32_t* item_idx, int64_t E, int I, int32_t* count /*[I], accumulated: zero it first*/, void* stream);
int dmm_graph_build_shard(const int32_t* user_ptr, const int32_t* edge_user, const int32_t* item_idx, int U, int I, int64_t E, int D, int task_len,
                          const double* dinv_table, const int32_t* item_count_global, int self_mod, int self_rank, int32_t* rowptr, int32_t* col,
                          float* val, void* tasks, void* fix, int32_t* meta, void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Evaluation: Coach.testEpoch (Main.py:403-411).  score [rows, cols] holds usrEmbeds[users] @ itmEmbeds^T (dmm_gemm); every
 * train item of users[r] (CSR trn_ptr/trn_col over ALL users, int64) is set to -1e8 IN PLACE -- the reference's
 * `allPreds * (1 - trnMask) - trnMask * 1e8` -- and the k best columns of the row are written to out[r*k .. r*k+k) in
 * descending score order (equal scores: smaller column first).  k <= 64.  One workgroup per row. */
int dmm_eval_topk(float* score, int64_t ld, const int32_t* users, const int64_t* trn_ptr, const int64_t* trn_col, int rows, int cols, int k,
                  int32_t* out, void* stream);

/* Coach.calcRes (Main.py:422-448) on the device: out[r*3 + {0,1,2}] = recall, ndcg, precision (fp64) of row r's k
 * recommendations top[r*k ..] against the TEST items of users[r] (CSR tst_ptr/tst_col over all users, int64).  Users
 * without test items get zeros.  The caller sums the rows (the reference sums them in Python). */
int dmm_eval_metrics(const int32_t* top, const int32_t* users, const int64_t* tst_ptr, const int64_t* tst_col, int rows, int k, double* out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
