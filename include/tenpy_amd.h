/* tenpy_amd.h -- C-ABI of the MI355X (gfx950) backend for TeNPy's block-sparse hot path.
 *
 * Every entry point replaces one native call site of the reference
 * (tenpy/linalg/_npc_helper.pyx, tenpy/linalg/np_conserved.py; cited per function).
 * Conventions:
 *   - all pointers named *_dev / *base are DEVICE pointers (HBM); `stream` is a hipStream_t
 *     passed as void* (NULL = default stream); everything is asynchronous on that stream
 *     unless the doc says "synchronises".
 *   - dtype: TPA_F64 (real double) or TPA_C128 (interleaved complex double) -- the only two
 *     calculation dtypes of the reference (_npc_helper.pyx:410-424 `_find_calc_dtype`).
 *   - block data: an Array's blocks are packed back to back in ONE device arena; a block is
 *     addressed as (arena base, element offset).  Offsets / sizes are in ELEMENTS of dtype.
 *   - return value: 0 = ok, >0 = hipError_t of the failing runtime call,
 *     <0 = TPA_E_* argument/algorithm error (mapped to ValueError / RuntimeError /
 *     LinAlgError by the Python shim, SURVEY 8(b) "Errors").
 */
#ifndef TENPY_AMD_H
#define TENPY_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TPA_F64 0
#define TPA_C128 1

#define TPA_E_BADARG (-1)      /* -> ValueError  */
#define TPA_E_NOCONV (-2)      /* -> numpy.linalg.LinAlgError (Jacobi sweeps exhausted)  */
#define TPA_E_NAN (-3)         /* -> ValueError("NaN ...") like np_conserved.py:4978-4982 */
#define TPA_E_NOMEM (-4)
#define TPA_E_RANKCAP (-5)     /* tpa_svd_batch with a rank cap: a block is not of low rank (caller falls back) */

/* ---- library / device --------------------------------------------------------------- */
int tpa_version(void);
/* Fills name (<=255 chars), number of CUs, HBM bytes of the current device. */
int tpa_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes);
const char *tpa_last_error(void);

/* ---- K1: grouped, chained GEMM  (replaces CblasGemmBatch.run, _npc_helper.pyx:204-273;
 *          python twin `fast_dot_sum`, np_conserved.py:4807-4837) ----------------------
 * The reference queues (A,B,C,m,k,n) per accumulation *level* and runs level 0 with beta=0
 * and later levels with beta=1.  Here the levels of one C block form a *chain* that is
 * accumulated in MFMA registers (no beta=1 re-read of C):
 *        C_t (m x n, row stride ldc)  =  [C_t +]  sum_{l in chain(t)} A_l (m x k_l) * B_l (k_l x n)
 * links  : int64[n_links][8]  = {a_off, b_off, k, a_rs, a_ks, b_ks, b_ns, flags}
 *          A_l(i,kk) = Abase[a_off + i*a_rs + kk*a_ks], B_l(kk,j) = Bbase[b_off + kk*b_ks + j*b_ns]
 *          (one of a_rs/a_ks and one of b_ks/b_ns should be 1 for coalescing; any strides are legal)
 *          flags bit0: conj(A), bit1: conj(B)   (complex only)
 * tasks  : int64[n_tasks][8]  = {c_off, m, n, ldc, link_begin, link_count, accumulate, 0}
 * tiles  : int32[n_tiles][4]  = {task, tile_row, tile_col, 0}   (tile shape: tpa_gemm_tile_shape(dtype, cfg))
 * cfg    : 0 = large tiles (128 x 128 real), 1 = small tiles (64 x 64 real) -- chosen per plan by the host
 *          so that the launch has enough workgroups for 256 CUs.
 * All three tables live on the DEVICE (uploaded once per cached contraction plan).
 */
int tpa_gemm_chain(int dtype, int cfg, const int64_t *tasks_dev, const int64_t *links_dev,
                   const int32_t *tiles_dev, int n_tiles, const void *Abase, const void *Bbase,
                   void *Cbase, void *stream);

/* ---- K2-K4: vector kernels over packed arenas (replace ddot/zdotc/zdotu
 *      _npc_helper.pyx:1854-1871, daxpy/zaxpy :316-336, dscal/zscal :339-364,
 *      np.linalg.norm per block np_conserved.py:2252-2255) ------------------------------
 * `n` counts elements of dtype.  Scalars alpha are passed as (re, im); im ignored for F64.
 * Reductions are deterministic two-pass (per-workgroup partials, then one workgroup);
 * `scratch_dev` must hold >= TPA_RED_SCRATCH doubles; the result (2 doubles: re, im) is
 * written to out_dev[0..1] on the stream (no host sync).
 */
#define TPA_RED_SCRATCH 4096
int tpa_axpy(int dtype, int64_t n, double alpha_re, double alpha_im, const void *x_dev,
             void *y_dev, void *stream);
int tpa_scal(int dtype, int64_t n, double alpha_re, double alpha_im, void *x_dev, void *stream);
/* out = sum conj?(x_i) * y_i ;  do_conj as in _inner_worker(a, b, do_conj) */
int tpa_dot(int dtype, int64_t n, const void *x_dev, const void *y_dev, int do_conj,
            double *out_dev, double *scratch_dev, void *stream);
/* out[0] = sum |x_i|^2  (2-norm squared; host takes the sqrt) */
int tpa_nrm2sq(int dtype, int64_t n, const void *x_dev, double *out_dev, double *scratch_dev,
               void *stream);
/* Fused Lanczos recurrence step (krylov_based.py:660-672):
 *   w -= alpha * v1 ; w -= beta * v0 (v0 may be NULL) ; out[0] = sum |w_i|^2           */
int tpa_lanczos_update(int dtype, int64_t n, void *w_dev, double alpha_re, double alpha_im,
                       const void *v1_dev, double beta_re, double beta_im, const void *v0_dev,
                       double *out_dev, double *scratch_dev, void *stream);
/* The same step with the scalars kept on the device, so that the host never waits between a matvec and the next one
 * (LanczosGroundState._build_krylov, krylov_based.py:655-672; the host reads alpha / beta one step late for the tridiagonal
 * eigen-problem and the convergence test):
 *   alpha = Re <w|v1> ;  w -= alpha v1 ;  w -= sqrt(bsq_prev[0]) v0 (v0, bsq_prev may be NULL) ;  bsq = |w|^2 ;  w /= sqrt(bsq)
 *   ab_out[0] = alpha, ab_out[1] = bsq.  scratch_dev: TPA_RED_SCRATCH doubles. */
int tpa_lanczos_step(int dtype, int64_t n, void *w_dev, const void *v1_dev, const void *v0_dev,
                     const double *bsq_prev_dev, double *ab_out_dev, double *scratch_dev, void *stream);

/* LanczosGroundState.run as ONE host call (krylov_based.py:645-700 `_build_krylov`; the caller keeps the reference's host
 * logic -- tridiagonal eigh :702, `_converged` :713 -- in `cb`).  The matvec is a replayed "program" of cached launches:
 *   ops : HOST int64[n_ops][12] = {kind, cfg, p0, p1, p2, count, a_slot, b_slot, c_slot, max_elems, 0, 0}
 *         kind 0: tpa_gemm_chain(dtype, cfg, tasks = p0, links = p1, tiles = p2, n_tiles = count, A, B, C)
 *         kind 1: tpa_lincomb_batch(dtype, jobs = p0, n_jobs = count, terms = p1, max_elems, src = A, dst = C); cfg = 1 marks the
 *                 reduction of the split-K partial blocks of the kind-0 op before it (timed together with that GEMM)
 *         kind 2: tpa_copy_batch(dtype, jobs = p0, n_jobs = count, max_elems, src = A, dst = C)   (pack / unpack of row panels)
 *         kind 3: the caller's collective number `cfg` -- `tpa_lanczos_set_collective` -- is invoked on the host at this point of
 *                 the program; it enqueues an exchange (RCCL all-gather of the row panels of a sharded matvec) that is
 *                 ordered on the launch stream.  The recurrence, its scalars and the stopping test stay replicated and bit-identical
 *                 on every rank, so all ranks run the same number of steps.
 *         slots: >= 0 -> bufs[slot] (HOST array of n_bufs device pointers: fixed operands and temporaries), -1 -> the input
 *         vector v_k, -2 -> the output vector w of this matvec.  (p0..p2 are device pointers stored as integers.)
 *   krylov_dev : (N_max + 1) * n elements; on return vectors 0 .. N-1 are the orthonormal Krylov basis (v_0 = psi0 / |psi0|).
 *   scalars_dev: 2 * (N_max + 2) doubles.  Per step k:  w = matvec(v_k) [+ E_shift v_k];  tpa_lanczos_step;  (alpha_k, beta_k^2)
 *   are posted to mapped host memory and `cb(k, alpha_k, beta_k^2, user)` is called ONE STEP LATE, while step k + 1 already runs
 *   on the device; a nonzero return stops the iteration after step k (N = k + 1; the step in flight is discarded).
 *   info (HOST double[4]) = {N, number of matvecs launched, summed GEMM time in ms if time_gemms else 0, |psi0|};
 *   N = 0 <=> |psi0| < cutoff (nothing useful was computed).                                                          */
typedef int (*tpa_lanczos_callback)(int step, double alpha, double beta_sq, void *user);
/* Collective hook of op kind 3 (per host thread; NULL = none): returns 0 on success. */
typedef int (*tpa_collective_callback)(int which, void *user);
int tpa_lanczos_set_collective(tpa_collective_callback cb, void *user);
int tpa_lanczos_run(int dtype, int64_t n, const int64_t *ops, int n_ops, void *const *bufs, int n_bufs,
                    void *krylov_dev, const void *psi0_dev, int N_max, double cutoff, int has_shift, double E_shift,
                    double *scalars_dev, double *scratch_dev, tpa_lanczos_callback cb, void *user,
                    int time_gemms, double *info, void *stream);
/* out = sum_{k < N} coeff[k] v_k over the Krylov basis of tpa_lanczos_run (N <= 64, real coefficients: the eigenvector of
 * the tridiagonal matrix), norm_host[0] = |out| (blocking): `_calc_result_full`, krylov_based.py:223-236, in one pass. */
int tpa_krylov_combine(int dtype, int64_t n, const void *krylov_dev, int N, const double *coeff, void *out_dev,
                       double *red_out_dev, double *scratch_dev, double *norm_host, void *stream);

/* ---- K8/K9/K10: data movement ---------------------------------------------------------
 * Generic strided N-d block copy (N <= TPA_COPY_MAXDIM), batched.  Replaces
 * _sliced_strided_copy/_sliced_copy (_npc_helper.pyx:368, :754; combine/split legs :1112-1123,
 * :1235-1240) and the per-block PyArray_Transpose+copy of itranspose (:853).
 * jobs : int64[n_jobs][4 + 3*TPA_COPY_MAXDIM] =
 *        {dst_off, src_off, ndim, flags, shape[MAXDIM], dst_stride[MAXDIM], src_stride[MAXDIM]}
 *        flags bit0: conjugate while copying (complex only).
 * The last dim is the fastest-varying loop index; strides in elements.
 */
#define TPA_COPY_MAXDIM 6
int tpa_copy_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems,
                   const void *src_base, void *dst_base, void *stream);
/* dst slab (rows x cols, row stride dst_ld) = sum_t alpha_t * src_t slab (row stride src_ld_t), batched.
 * Fuses tensordot(LP, W) + combine_legs of MPOEnvironment._contract_LHeff / _contract_RHeff (networks/mpo.py:3107,
 * :3118; TwoSiteH.combine_Heff, mps_common.py:1350) when every charge block of the MPO tensor W is a single number:
 * each (w', p, p*) slab of LHeff is a linear combination of the LP[:, w, :] blocks with the W entries as coefficients.
 * jobs : int64[n_jobs][8] = {dst_off, rows, cols, dst_ld, term_begin, term_count, 0, 0}
 * terms: int64[n_terms][4] = {src_off, src_ld, alpha_re, alpha_im}  (alpha_* are the IEEE-754 bit patterns of doubles) */
int tpa_lincomb_batch(int dtype, const int64_t *jobs_dev, int n_jobs, const int64_t *terms_dev, int64_t max_job_elems,
                      const void *src_base, void *dst_base, void *stream);
/* x_b[i, j, l] *= s[s_off_b + j]  for each block b viewed as (pre, len, post).  Replaces
 * iscale_axis, np_conserved.py:2132-2140.  jobs: int64[n][6] = {x_off, pre, len, post, s_off, 0};
 * the scale vector s is real (F64) or of `dtype` when s_is_complex. */
int tpa_scale_axis_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems,
                         void *x_base, const void *s_dev, int s_is_complex, void *stream);
int tpa_fill_zero(void *dst_dev, int64_t n_bytes, void *stream);
/* G_b <- strict lower triangle of G_b, diagonal (G_ii - 1) / 2, zeros above, for square blocks G_b (n x n row-major at g_off).
 * With G = T T^H the Gram matrix of row vectors sorted by descending singular value, T <- T - G T orthonormalises every vector
 * against the vectors before it (one ordered Gram-Schmidt step on the matrix cores, defect d -> O(d^2)): the post-processing of
 * the singular vectors below the absolute floor of the block SVD, where the reference gets LAPACK's orthonormal vectors
 * (np_conserved.py:4970 svd_flat).  jobs: int64[n][2] = {g_off, n}. */
int tpa_tri_lower_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems, void *g_base, void *stream);
/* dst[b][i, j, l] = src[b][i, idx[idx_off + j], l] for blocks viewed as (pre, len, post): the np.compress
 * of iproject (np_conserved.py:1982).  jobs: int64[n][8] = {dst_off, src_off, pre, len_src, len_dst, post,
 * idx_off, 0}; idx_dev: int64 indices. */
int tpa_gather_axis_batch(int dtype, const int64_t *jobs_dev, int n_jobs, int64_t max_job_elems,
                          const int64_t *idx_dev, const void *src_base, void *dst_base, void *stream);
/* out[o_off + j] = sum_{i,l} |x[i, j, l]|^2 for blocks viewed as (pre, len, post): per-slice squared norms
 * (np.linalg.norm(block, axis) in _qr_theta_Y0, truncation.py:452).  jobs: int64[n][6] = {x_off, pre, len, post,
 * o_off, 0}; rows: int32[n_rows][2] = {job, j} (one wavefront each; n_rows padded to a multiple of 4 with job=-1). */
int tpa_axis_sqnorm_batch(int dtype, const int64_t *jobs_dev, const int32_t *rows_dev, int n_rows,
                          const void *x_base, double *out_dev, void *stream);
/* Flat dtype conversion / (complex) conjugation of an arena: astype (np_conserved.py:1865) and
 * iconj's complex_conj (np_conserved.py:2202-2235).  c128->f64 keeps the real part. */
int tpa_convert(int from_dtype, int to_dtype, int64_t n, const void *src_dev, void *dst_dev, int conj,
                void *stream);
/* Tile shape (rows, cols of C per workgroup) of GEMM configuration `cfg` for `dtype`. */
int tpa_gemm_tile_shape(int dtype, int cfg, int *bm, int *bn);
/* Tuning hook (also TPA_GEMM_VARIANT in the environment): bit 0 = the round-6 loop of the real kernel also for launches of <= 1024
 * tiles (default there: eight wavefronts per 64 x 64 tile on the round-4 loop; profiles/r06_gemm_v2.txt). */
int tpa_gemm_set_variant(int v);

/* ---- K5: batched block SVD, one-sided (Hestenes) Jacobi -- replaces svd_flat / LAPACK gesdd
 *      per charge block (np_conserved.py:4970-4980 via svd_robust.py:36-75) ---------------
 * jobs : int64[n_jobs][8] = {a_off, m, n, u_off, s_off, vh_off, flags, norm2_bits}  (HOST pointer)
 *   norm2_bits: 0, or the IEEE-754 bit pattern of a double that replaces |A_b|_F^2 as the scale of the numerical-rank
 *   decision and of the absolute floor (used for residual blocks E = A - P whose own norm is far below that of A).
 *   flags bit 0 (square blocks only): orthogonalise the rows of A instead of its columns (default 0).
 *   A_b is m x n row-major at a_off in a_base; on return
 *   U_b (m x k, row-major, k=min(m,n)) at u_off in u_base, S_b (k, descending) at s_off in s_dev
 *   (always real), VH_b (k x n, row-major) at vh_off in vh_base.  A is NOT overwritten.
 * work_dev: >= tpa_svd_worksize(...) bytes.  Synchronises the stream (sweep-convergence test).
 * min(m,n) >= 32 (complex: and max(m,n) <= 2048): the blocks are first reduced by a rank-revealing Householder QR with column
 *   pivoting (X P = Q [R;0], X = A or A^T); the Jacobi iteration then runs on the r x min(m,n) factor only
 *   (r = numerical rank: residual column norms <= 1e-15 ||A||_F).  Singular values below that threshold are
 *   returned as exact zeros with zero singular vectors (LAPACK returns rounding noise there).
 * tol: absolute floor rho of the stopping rule, |tol| <= 1 (0: the purely relative Hestenes criterion, every pair converged to a
 *   cosine of eps sqrt(L)).  tol > 0: pairs whose LARGER row is below rho |A|_F stop at |x.y| <= eps sqrt(L) |y| rho |A|_F (rounds
 *   1 - 5; exact after a SYMMETRIC re-orthonormalisation of the small vectors).  tol < 0 (round 6, what the engines pass): the floor
 *   |tol| acts on the SMALLER row, |x.y| <= eps sqrt(L) |x| max(|y|, rho |A|_F) with |x| >= |y| -- the same absolute accuracy of
 *   U S VH and of S provided the caller orthonormalises the normalised rows in DESCENDING order of S (every vector against the
 *   ones before it: tpa_tri_lower_batch), far fewer rotations on spectra graded down to rounding level.
 * Returns TPA_E_NOCONV if max_sweeps is exhausted, TPA_E_NAN if the input holds NaN/Inf.
 */
int64_t tpa_svd_worksize(int dtype, const int64_t *jobs_host, int n_jobs);
int tpa_svd_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                  void *u_base, double *s_dev, void *vh_base, void *work_dev, int64_t work_bytes,
                  int max_sweeps, double tol, int *sweeps_done, void *stream);

/* ---- the per-bond SVD section of a sweep as ONE call (round 6) -- replaces, for a two-site wave function whose bond has been
 *      decomposed before, the per-block LAPACK calls of npc.svd (np_conserved.py:3676-3760, worker :4950-5002) inside svd_theta
 *      (truncation.py:258): the WARM route (project on the singular vectors Bq of the bond's previous visit, residual test, one-sided
 *      Jacobi on W = Bq X^H without any QR, accumulated basis, results in the layout of tpa_svd_batch, singular values on the host,
 *      ordered clean-up of the vectors below the absolute floor).  All tables are built inside and uploaded in one copy per stage.
 * side   : 0 = 'R' (X = A: the basis spans the row space of theta; sweep moving right), 1 = 'L' (X = A^T).
 * blocks : HOST int64[n_blocks][8] = {a_off, m, n, u_off, s_off, v_off, b_off, b_k}: A_b (m x n row-major at a_off; the blocks lie back
 *          to back and fill a_numel elements), results U_b (m x kk) / S_b (kk) / VH_b (kk x n) at u_off / s_off / v_off (kk = min(m, n)),
 *          basis Bq_b (b_k x n (side 0) or b_k x m (side 1), row-major, orthonormal rows, 0 < b_k <= kk) at b_off in basis_arena.
 * u_arena / v_arena (u_numel / v_numel elements) are cleared here; s_host (HOST, sum of kk doubles) receives S (zero beyond b_k).
 * e_tol  : per block |X - (X Bq^H) Bq|_F <= e_tol |X|_F or the call returns 1 ("stale basis") with info[0] = the worst relative
 *          residual, info[1] = number of failing blocks, and the result arenas cleared -- the caller goes on with the sketch / cold route.
 * lowdin_basis: one first-order Loewdin step on the accumulated basis (every few warm generations of a bond).
 * clean_iterations, clean_floor: ordered re-orthonormalisation (tpa_tri_lower_batch) of the normalised Jacobi rows over the
 *          significant vectors when some lie below clean_floor |S_b|_2 (0 iterations: none) -- the counterpart of tol < 0 below.
 * alg_warm / alg_restore: tpa_svd_set_algorithm values for the Jacobi stage (bit 9: no pivoted QR) and afterwards.
 * max_sweeps, tol, sweeps_done: as for tpa_svd_batch.  info: HOST double[4].  f64 only.  Waits for the residual test and for the
 * singular values; the result copies and the clean-up may still be queued on the stream when it returns.
 * Returns 0 (done), 1 (stale), or a TPA_E_* code of the Jacobi stage. */
int tpa_svd_theta(int dtype, int side, const int64_t *blocks, int n_blocks, int64_t a_numel, const void *a_arena,
                  const void *basis_arena, void *u_arena, int64_t u_numel, void *v_arena, int64_t v_numel, double *s_host,
                  double e_tol, int lowdin_basis, int clean_iterations, double clean_floor, int alg_warm, int alg_restore,
                  int max_sweeps, double tol, int *sweeps_done, double *info, void *stream);
/* The warm-start bases of the bond's next visit out of a finished decomposition: rows [0, ksig_b) of VH_b -> basis_r (ksig x n,
 * back to back), columns [0, ksig_b) of U_b transposed -> basis_l (ksig x m).  blocks as above (m, n, u_off, v_off are read). */
int tpa_svd_theta_store(int dtype, const int64_t *blocks, const int64_t *ksig, int n_blocks, const void *u_arena, const void *v_arena,
                        void *basis_r, void *basis_l, void *stream);

/* Algorithm switch (test / benchmark hook): 0 (default) = pivoted-QR preconditioner + block Jacobi (16-row MFMA
 * Gram + in-LDS eigen-solve); bit 0 = one wavefront per row pair; bit 1 = two-kernel Jacobi rounds instead of the fused one; bit 2 = full local
 * sweep in every round; bits 4-7 = local sweeps; bit 9 (512) = no pivoted-QR preconditioner; bit 10 (1024) = NO predicted
 * convergence; bit 11 (2048) = no one-workgroup-per-pair round (real data, blocks with rank + columns <= 2048; the fused round with
 * column parts is used instead); bit 12 (4096) = no 32-row-block rounds (real data; the 8-row-block kernels run instead); bit 13 (8192) = no
 * look-ahead round (the host drains the stream after every sweep before it enqueues the next one).  Predicted convergence (default since round 2, validated on the MI355X on chi = 2048 blocks: same singular values
 * to 1.4e-15 sigma_max, same orthogonality, one to two sweeps fewer): a sweep in which no rotated pair had a scaled cosine
 * above 1e-7 ends the iteration without the verification sweep (quadratic convergence leaves cosines <= 1e-14).
 * Round 4 -- Gram-only sweeps (csrc/tpa_svd_b32.inc, real data; csrc/tpa_svd_b32c.inc, complex data -- there they are the only
 * 32-row-block path; calls whose largest block has >= 96 rows; default): a sweep starts
 * from ONE exact Gram matrix W W^T per block (grouped GEMM), its rounds rotate that matrix alone (solve per pair + 64^3 MFMA
 * updates of the Gram tiles and of the accumulated transform) and end with one product [W | G] <- Qtot [W | G]; bit 20
 * (1048576) = off (gram / solve / apply on the data in every round, the round-3 path; complex data: the 8-row-block rounds).
 * Bit 14 (16384): complex Gram-only rounds with the tiles the next solve does not read on a second stream (measured slower; off).
 * Bit 23 (8388608): two launches per Gram-only round (round-4 A/B switch; bit 22 selected the round-3 solve kernel, removed in round 5).
 * Round 5: the "big rotation" test of the predicted convergence judges a pair below the floor against its own stopping rule
 * (scale sqrt(max(alpha, beta)) rho |A|_F instead of rho^2 |A|_F^2: with the old scale such pairs never counted and the iteration
 * could end on cosines of O(0.1) among them).  (Bits 15 - 19 and 21 switched the round-4 end game by simultaneous rotations +
 * Newton-Schulz steps: measured slower over a whole sweep, never on by default, removed in round 5 with its counters.) */
int tpa_svd_set_algorithm(int pairwise);
/* Round 6 -- activity-driven rounds of the Gram-only sweeps (real data; bit 24 (16777216) of tpa_svd_set_algorithm = off): at the start
 * of a sweep a kernel tests, on the exact Gram matrix, which pairs of 32-row blocks contain a row pair that needs a rotation; the host
 * packs only those block pairs into perfect matchings (the rounds of that sweep) and ends the iteration after a sweep that started
 * without "big" pairs.  tpa_svd_dyn_stats: out = {rounds launched, rounds the round-robin schedule would have run, sweeps}. */
int tpa_svd_dyn_stats(int64_t *out, int reset);
/* Diagnostic ring of the most recent tpa_svd_batch calls (host): rows of out = {min(m, n) of the largest block, its max(m, n),
 * blocks, sweeps, pivoted QR used, algorithm switches, wall microseconds inside the call, return code}; returns the rows written. */
int64_t tpa_svd_call_log(int64_t *out, int64_t max_rows, int reset);
/* Rank cap of the pivoted-QR stage (0 = none, default): with cap > 0 tpa_svd_batch returns TPA_E_RANKCAP as soon as some
 * block turns out to have numerical rank above ~cap (checked every 64 columns).  Used by the warm-started SVD for the residual
 * blocks E = A - P, which are decomposed only if they are of low rank (tenpy_amd/linalg/_svd_warm.py). */
int tpa_svd_set_rank_cap(int cap);

/* ---- K6: batched Householder QR (np.linalg.qr per block, np_conserved.py:4190) ----------
 * jobs : int64[n_jobs][8] = {a_off, m, n, q_off, r_off, 0,0,0} (HOST); reduced mode:
 *   Q_b m x k row-major, R_b k x n row-major, k = min(m,n).  A not overwritten.
 *   a_off is a signed element offset from a_base: the blocks of SEVERAL arenas may be factorised in one call by taking the
 *   lowest arena address as a_base (np_conserved.qr_batched: the bonds of one half-step of the QR-based TEBD, reference
 *   algorithms/tebd.py:374-414 / truncation.py:611-640). */
int tpa_qr_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base, void *q_base,
                 void *r_base, void *stream);
/* Test hook: bit 0 = always use the one-workgroup Householder kernel (default: blocked compact-WY QR on the matrix
 * cores when min(m,n) >= 32 and m <= 8192 (real) / 2048 (complex)); bit 1 = two launches per 8-column panel (factorisation, then
 * trailing update: rounds 2-4) also where the default is ONE launch per panel (round 5, csrc/tpa_qr_la.inc: real data, every block
 * tall with <= 2048 rows -- workgroup 0 of a block brings panel k + 1 up to date and factorises it while the other workgroups apply
 * panel k behind it).  Since round 5 tpa_qr_batch also orthogonalises the sketch Y = X Omega^H of the warm-started block SVD
 * (linalg/_svd_warm.py::svd_blocks_sketch). */
int tpa_qr_set_algorithm(int v);

/* ---- K7: batched Hermitian eigendecomposition (np.linalg.eigh per block, np_conserved.py:5059-5061; `_eig_worker` :5041)
 * jobs : int64[n_jobs][8] = {a_off, n, w_off, v_off, 0,0,0,0} (HOST);  A_b n x n Hermitian row-major (lower triangle read, UPLO = 'L'),
 *   eigenvalues ascending at w_off in w_dev, eigenvectors as COLUMNS of V_b (n x n row-major).
 * Blocks of >= 96 rows: TWO-SIDED block Jacobi on A_b + mu (mu = 2 |A_b|_F) itself -- 32-row blocks, per round one cyclic Jacobi solve
 * of every 64 x 64 diagonal pair block and the two-sided MFMA update S[P, P'] <- Q_P S[P, P'] Q_P'^H, Qtot[P, :] <- Q_P Qtot[P, :];
 * no GEMM over the data, no squared spectrum; stops when no |S_ij| > eps sqrt(n) sqrt(S_ii S_jj) is left (absolute accuracy
 * eps sqrt(n) |A_b|_F, LAPACK's class).  Smaller blocks (and tpa_eigh_set_direct(0)): the one-sided iteration on the rows of A_b + mu.
 * All blocks of the call share the launches: callers batch independent matrices (np_conserved.eigh_batched: the bond matrices of a
 * TEBD half-step).  work_dev >= tpa_eigh_worksize bytes.  Synchronises the stream. */
int64_t tpa_eigh_worksize(int dtype, const int64_t *jobs_host, int n_jobs);
int tpa_eigh_batch(int dtype, const int64_t *jobs_host, int n_jobs, const void *a_base,
                   double *w_dev, void *v_base, void *work_dev, int64_t work_bytes, int max_sweeps,
                   double tol, int *sweeps_done, void *stream);
/* Test hook: 0 = tpa_eigh_batch always takes the shift + one-sided route (rounds 1 - 5), 1 = default. */
int tpa_eigh_set_direct(int on);
/* Eigenpairs of Hermitian blocks out of their SVDs A_b = U_b S_b VH_b (tpa_svd_batch), WITH the check that this is legitimate: where |lambda|
 * is not shared by a positive and a negative eigenvalue, v_i = d_i u_i (d_i = +/-1), lambda_i = d_i S_i and u_i is the eigenvector.
 * np_conserved.eigh_batched takes this route for real data (the graded, rank-deficient density matrices of the mixer, mps_common.py:1972-2079:
 * ~7 Jacobi sweeps on the rank-r factor instead of 35 - 40 on the matrix) and falls back to tpa_eigh_batch if err is not at rounding level.
 * jobs : int64[n_jobs][8] = {u_off, n, s_off, vh_off, lam_off, 0,0,0} (HOST);  lam_dev[lam_off + i] = d_i S_i (order of S: descending |lambda|),
 * err_dev[job] = max_i S_i |v_i - d_i u_i| = max_i |A u_i - lambda_i u_i|.  Asynchronous on `stream`. */
int tpa_eigh_from_svd(int dtype, const int64_t *jobs_host, int n_jobs, const void *u_base, const double *s_dev,
                      const void *vh_base, double *lam_dev, double *err_dev, void *stream);

/* ---- host planner: integer bookkeeping of _tensordot_worker (_npc_helper.pyx:1498-1786) ---
 * Inputs describe operand a with its contracted legs LAST and b with its contracted legs FIRST
 * (i.e. after _tensordot_transpose_axes, :1260-1295), block lists in any order.
 *   a_qdata : int64[na][ra]  qindices,  a_keep = ra - ncontr ;  likewise b.
 *   *_block_size: for each leg, `nblk` block sizes are looked up through leg_sizes/leg_ptr:
 *       size of qindex q on leg L of a = a_leg_sizes[a_leg_ptr[L] + q].
 * Outputs (caller-allocated, capacities given; counts returned through n_*):
 *   res_qdata   : int64[n_res][a_keep + b_keep]   lex-sorted like the reference (:1777)
 *   res_a_first, res_b_first: representative a / b block of each result block
 *   gemm        : int64[n_gemm][3] = {res_index, a_block, b_block}  ordered by (res_index, level)
 * Returns 0, or TPA_E_BADARG if a capacity is too small (n_* then hold the needed sizes).
 */
int tpa_plan_tensordot(const int64_t *a_qdata, int64_t na, int ra, const int64_t *b_qdata,
                       int64_t nb, int rb, int ncontr, const int64_t *contr_nblocks,
                       int64_t *res_qdata, int64_t *res_a_first, int64_t *res_b_first,
                       int64_t cap_res, int64_t *n_res, int64_t *gemm, int64_t cap_gemm,
                       int64_t *n_gemm);

#ifdef __cplusplus
}
#endif
#endif
