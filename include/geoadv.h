/*
 * geoadv.h -- C ABI of libgeoadv.so, the MI355X (gfx950) implementation of the geometric
 * adversarial attack hot path of itailang/geometric_adv.
 *
 * This header is the drop-in boundary.  Every entry point below replaces one native launcher
 * (or one session-level Python method) of the reference; the reference file:line is cited at
 * each declaration.  Conventions, all entry points:
 *   - plain C linkage, plain pointers and ints; no torch / TF types,
 *   - every data pointer is a DEVICE pointer (HBM) unless the parameter name starts with host_,
 *   - row-major contiguous fp32 / int32 tensors in the reference's layouts,
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only ENQUEUE
 *     work (no allocation, no host synchronisation) unless stated otherwise, so they are
 *     capturable into a hipGraph,
 *   - return value: 0 on success, non-zero on error; geoadv_last_error() gives the message
 *     (thread-local).  Shape violations that the reference reports through OP_REQUIRES /
 *     errors::InvalidArgument (tf_nndistance.cpp:51-58, tf_grouping.cpp:70-74) return
 *     GEOADV_EINVAL here,
 *   - the caller owns every buffer; the library owns only what lives behind the opaque
 *     geoadv_ae / geoadv_attack handles.
 */
#ifndef GEOADV_H
#define GEOADV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOADV_OK      0
#define GEOADV_EINVAL  1   /* bad shape / argument                      */
#define GEOADV_EHIP    2   /* a HIP runtime call or kernel launch failed */
#define GEOADV_ENOMEM  3
#define GEOADV_ERANGE  4   /* a value left the range an arithmetic was set up for (geoadv_ae_status) */

int         geoadv_version(void);                /* 1000*major + minor */
const char *geoadv_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Structural losses: external/structural_losses
 * ---------------------------------------------------------------------------------------- */

/* NmDistanceKernelLauncher(b,n,xyz,m,xyz2,result,result_i,result2,result2_i)
 * (tf_nndistance.cpp:168, kernel tf_nndistance_g.cu:5-131).  dist1[b,n], idx1[b,n] = squared
 * distance / index of the nearest xyz2 point for every xyz1 point; dist2/idx2 the converse.
 * Results are bit-identical to the reference CPU op (tf_nndistance.cpp:21-43): unfused fp32
 * arithmetic, lowest index on ties.  n == 0 or m == 0 is accepted (m == 0 gives dist 0 / idx 0,
 * like the CPU loop).  Non-finite coordinates follow the CPU loop too (`k==0 || d<best`, :33): a
 * query whose distance to candidate 0 is NaN returns (NaN, 0), a NaN distance to a later
 * candidate never wins, an infinite minimum returns its lowest index (the payload of a returned
 * NaN is not specified): a cloud pair with a non-finite coordinate is recomputed by that loop, a
 * thread per query, in a closing launch of the operator.  The same holds for geoadv_nn_distance_sym
 * and geoadv_nn_distance_paired (tests/golden/nn_distance_nonfinite.npz). */
int geoadv_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2,
                       float *dist1, int *idx1, float *dist2, int *idx2, void *stream);

/* The same outputs, bit for bit, from ONE evaluation of every pair distance for both directions (the attack loop's kernel,
 * csrc/chamfer_sym.hip: the distance matrix is symmetric in the two clouds' roles down to the last bit).  n, m >= 1.  workspace:
 * geoadv_nn_distance_sym_workspace_floats(b,n,m) floats of caller-owned scratch (row minima per column slice; the caller
 * allocates, as with the temp tensor of tf_approxmatch.cpp:164-170). */
size_t geoadv_nn_distance_sym_workspace_floats(int b, int n, int m);
/* 1 if this shape is answered by the matrix-pipe-screened form of the scan (csrc/chamfer_mx.h: approximate distances from one fp16
 * MFMA per 32 x 32 pairs select, with a rigorous error bound, the few pairs that are then evaluated with the reference's
 * arithmetic -- same bits out), 0 if by the unscreened scan.  geoadv_set_chamfer_screen(0) turns the screened kernel off for the
 * whole process (operators, scorer, attack and training loops; returns the previous setting): the tests' second opinion. */
int geoadv_nn_distance_sym_is_screened(int b, int n, int m);
int geoadv_set_chamfer_screen(int on);
int geoadv_nn_distance_sym(int b, int n, const float *xyz1, int m, const float *xyz2,
                           float *dist1, int *idx1, float *dist2, int *idx2,
                           float *workspace, size_t workspace_floats, void *stream);

/* chamfer_dist[b] = reduce_mean(dist1[b,:]) + reduce_mean(dist2[b,:]) -- the scalar every caller of nn_distance forms next
 * (adv_ae.py:121,132; get_dists_per_point.py:75; prepare_indices_for_attack.py:114) -- in the summation order of the
 * attack loop's own metrics, so that a Chamfer distance recomputed from saved clouds equals adversarial_metrics[:,:,2]
 * bit for bit (the np.array_equal sanity check of get_dists_per_point.py:114-115). */
int geoadv_chamfer_per_pc(int b, int n, int m, const float *dist1, const float *dist2, float *out, void *stream);

/* Same results as geoadv_nn_distance for n == m <= 8192, from an exact grid search that uses xyz2[j] as the first guess
 * for the neighbour of xyz1[j] (and vice versa): fast when the clouds are paired like the attack's (adv, x), never wrong
 * otherwise (a query whose guess is poor is scanned against all points). */
int geoadv_nn_distance_paired(int b, int n, const float *xyz1, const float *xyz2,
                              float *dist1, int *idx1, float *dist2, int *idx2, void *stream);

/* NmDistanceGradKernelLauncher(b,n,xyz1,m,xyz2,grad_dist1,idx1,grad_dist2,idx2,grad_xyz1,grad_xyz2)
 * (tf_nndistance.cpp:208, kernel tf_nndistance_g.cu:132-157).  Outputs are fully overwritten
 * (the reference memsets them).  Unlike the reference GPU kernel (float atomicAdd) the
 * accumulation order is the CPU op's (tf_nndistance.cpp:126-163), so results are deterministic
 * and bit-identical to it.  Requires n, m <= 32768. */
int geoadv_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                            const float *grad_dist1, const int *idx1,
                            const float *grad_dist2, const int *idx2,
                            float *grad_xyz1, float *grad_xyz2, void *stream);

/* Bulk Chamfer scorer: the graph of attacker/prepare_indices_for_attack.py:110-114 evaluated for
 * EVERY pair of clouds, out[i*nb + j] = mean_p dist1 + mean_q dist2 of nn_distance(A[i], B[j]), without the
 * np.tile'd (na, nb, n, 3) copies the script feeds in 10 x 10 batches (:121-139).  A (na,n,3), B (nb,m,3).
 * workspace: geoadv_chamfer_matrix_workspace_floats(na,nb,n,m) floats (any larger or smaller size >= one pair's
 * worth works: pairs are processed in chunks that fit). */
size_t geoadv_chamfer_matrix_workspace_floats(int na, int nb, int n, int m);
int geoadv_chamfer_matrix(int na, int nb, int n, int m, const float *A, const float *B, float *out,
                          float *workspace, size_t workspace_floats, void *stream);

/* approxmatchLauncher(b,n,m,xyz1,xyz2,match,temp) (tf_approxmatch.cpp:141, kernel
 * tf_approxmatch_g.cu:1-181).  match is (b,m,n) like the reference GPU op: match[b,l,k] couples
 * xyz2 point l with xyz1 point k.  temp: scratch of geoadv_approx_match_temp_floats(b,n,m) floats
 * (the reference allocates b*(n+m)*2, tf_approxmatch.cpp:168).  The level schedule is the CPU
 * op's (11 levels, tf_approxmatch.cpp:31). */
size_t geoadv_approx_match_temp_floats(int b, int n, int m);
int geoadv_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2,
                        float *match, float *temp, void *stream);
/* How the pair weight expf(level * |p - q|^2) (tf_approxmatch.cpp:46) is evaluated; the rest of the algorithm (fp64
 * capacities, factors and sums) is the same in both modes (csrc/emd.hip header):
 *   FAST       fp32 distance, v_exp_f32.  Plan entries typically within 1e-6 relative of the CPU op; on clouds of
 *              thousands of points a few entries per million reach ~1e-4 (the algorithm amplifies the weight rounding).
 *   REFERENCE  the CPU op's own weights bit for bit (double distance, float expf argument, glibc's expf algorithm in fp64)
 *              and double level terms: every entry within ~2 float ulps of the CPU op; ~3.7x the time.
 * geoadv_approx_match / geoadv_emd_cost_grad1 are the FAST mode. */
#define GEOADV_EMD_FAST 0
#define GEOADV_EMD_REFERENCE 1
/* OR-ed into a `mode` argument (and into geoadv_attack_config.emd_weight_mode): every sweep of THIS call in its dense form -- the
 * first three levels' sweeps (level = -4^8, -4^7, -4^6: weights exactly 0 beyond 0.04 / 0.08 / 0.16) otherwise walk a cell grid
 * instead of every pair (csrc/emd.hip, "Sparse levels").  The results differ only in the order of fp64 additions. */
#define GEOADV_EMD_DENSE_LEVELS 0x100
int geoadv_approx_match_mode(int mode, int b, int n, int m, const float *xyz1, const float *xyz2,
                             float *match, float *temp, void *stream);
/* TEST / MEASUREMENT ONLY: the process default of calls whose mode carries no GEOADV_EMD_DENSE_LEVELS flag (on != 0, the default:
 * sparse first levels; 0: every sweep dense).  An atomic word read at launch time -- a process that wants one behaviour for one
 * call or handle passes the flag instead; this setter is not meant to be flipped while other threads launch. */
int geoadv_emd_sparse_levels(int on);
/* matchcostLauncher (tf_approxmatch.cpp:142; tf_approxmatch_g.cu:183-227): out[b].  The reference's launcher has no scratch
 * argument: this form takes stream-ordered scratch of its own (hipMallocAsync on `stream`); geoadv_match_cost_ws is the same op on
 * caller-owned scratch of geoadv_match_cost_workspace_floats(b,n,m) floats. */
int geoadv_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *out, void *stream);
size_t geoadv_match_cost_workspace_floats(int b, int n, int m);
int geoadv_match_cost_ws(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                         float *workspace, size_t workspace_floats, void *stream);
/* match_cost and match_cost_grad w.r.t. xyz1 of the plan approx_match(xyz1, xyz2) WITHOUT materialising the plan: what a
 * caller that treats the plan as a constant (ApproxMatch is NoGradient, tf_approxmatch.py:19) and differentiates w.r.t.
 * xyz1 only needs -- the attack loop's Chamfer+EMD loss.  cost[b], grad1[b,n,3]; temp: scratch of
 * geoadv_emd_cost_grad1_temp_floats(b,n,m) floats.  Equal to the three separate ops up to the order of the fp32 sums. */
size_t geoadv_emd_cost_grad1_temp_floats(int b, int n, int m);
int geoadv_emd_cost_grad1(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1,
                          float *temp, void *stream);
int geoadv_emd_cost_grad1_mode(int mode, int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                               float *grad1, float *temp, void *stream);
/* matchcostgradLauncher (tf_approxmatch.cpp:143; tf_approxmatch_g.cu:229-295).  grad2 may be NULL (only grad1 wanted). */
int geoadv_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                           const float *match, float *grad1, float *grad2, void *stream);

/* ------------------------------------------------------------------------------------------
 * Grouping: external/grouping
 * ---------------------------------------------------------------------------------------- */

/* queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt) (tf_grouping.cpp:66;
 * tf_grouping_g.cu:3-36,125-128). */
int geoadv_query_ball_point(int b, int n, int m, float radius, int nsample,
                            const float *xyz1, const float *xyz2, int *idx, int *pts_cnt, void *stream);
/* selectionSortLauncher(b,n,m,k,dist,outi,out) (tf_grouping.cpp:108; tf_grouping_g.cu:83-131).
 * dist, outi, out are (b,m,n).  The first k entries of every row equal the reference's; the
 * remaining n-k entries hold the unselected elements in the reference's (swap) order too. */
int geoadv_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream);
/* groupPointLauncher(b,n,c,m,nsample,points,idx,out) (tf_grouping.cpp:142; tf_grouping_g.cu:40-57). */
int geoadv_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                       float *out, void *stream);
/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points) (tf_grouping.cpp:173;
 * tf_grouping_g.cu:61-78).  grad_points is zeroed here (the reference op memsets it,
 * tf_grouping.cpp:204) and accumulated in a fixed order (no float atomics). */
int geoadv_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                            float *grad_points, void *stream);
/* ... on caller-owned scratch (geoadv_group_point_grad_workspace_bytes bytes) instead of the stream-ordered allocation the
 * reference-shaped form above makes (its launcher prototype has no scratch argument). */
size_t geoadv_group_point_grad_workspace_bytes(int b, int n, int c, int m, int nsample);
int geoadv_group_point_grad_ws(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                               float *grad_points, void *workspace, size_t workspace_bytes, void *stream);
/* knn_point(k, xyz1, xyz2) (tf_grouping.py:48-75) fused: squared distances + the k smallest per
 * query in the order SelectionSort produces (incl. its swap tie rule), without the (b,m,n)
 * matrix or the two tiled (b,m,n,3) operands.  val, idx are (b,m,k).  1 <= k <= 64. */
int geoadv_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2,
                     float *val, int *idx, void *stream);
/* defender/get_knn_dists_per_point.py:78-81 fused: knn_point(k+1, pc, pc), drop the first column,
 * gather, euclidean distance.  out is (b,n,k). */
int geoadv_knn_dists(int b, int n, int k, const float *pc, float *out, void *stream);
/* Which kernel answers a k-NN call with k <= 16 (same results, bit for bit): */
#define GEOADV_KNN_AUTO        0   /* by size: datasets of >= 512 points the exact grid search, smaller ones the all-points kernel */
#define GEOADV_KNN_ALL_POINTS  1
#define GEOADV_KNN_GRID        2   /* the grid search at every size                                                               */
#define GEOADV_KNN_GRID_SHELLS 3   /* as GRID without the lane-private first pass (every query by the wave-uniform shell walk)     */
/* The two ops with the kernel selected PER CALL (`kernel` = GEOADV_KNN_*) on caller-owned scratch of
 * geoadv_knn_workspace_bytes(b,n,m,k) bytes (knn_dists: m = n and k + 1 neighbours) -- no allocation, nothing process-wide: what
 * concurrent host threads call.  The reference-shaped forms above (tf_grouping.py:48-75 has no scratch argument) allocate
 * stream-ordered scratch per call and follow the process default below. */
size_t geoadv_knn_workspace_bytes(int b, int n, int m, int k);
int geoadv_knn_point_ws(int kernel, int b, int n, int m, int k, const float *xyz1, const float *xyz2,
                        float *val, int *idx, void *workspace, size_t workspace_bytes, void *stream);
int geoadv_knn_dists_ws(int kernel, int b, int n, int k, const float *pc, float *out,
                        void *workspace, size_t workspace_bytes, void *stream);
/* TEST / MEASUREMENT ONLY: the process default (GEOADV_KNN_*) of geoadv_knn_point / geoadv_knn_dists; an atomic word read at
 * launch time, not meant to be flipped while other threads launch -- those pass `kernel` to the _ws forms. */
int geoadv_knn_grid_mode(int mode);

/* get_outlier_pc_inlier_pc (src/adversary_utils.py:149-178) on the device, fused with the score its caller forms
 * (defender/run_defense_surface.py:187-191: the mean of the first top_k kNN distances of a point): a point is an outlier if
 * mean(knn_dists[b,p,0:top_k]) > thresh, an inlier if <= thresh (a NaN score is neither, as with np.where).  knn_dists is
 * (b,n,knn_stride), 1 <= top_k <= min(knn_stride, 7); with knn_stride = top_k = 1 it is the reference function's own
 * per-point scalar.  Outputs as the reference's: inlier_pc / outlier_pc (b,n,3) packed in point order and padded with the last
 * packed point (zeros if none), outlier_idx (b,n) int16 zero-padded, outlier_num (b) int16.  outlier_pc / outlier_idx /
 * outlier_num may be NULL. */
int geoadv_outlier_filter(int b, int n, const float *pc, const float *knn_dists, int knn_stride, int top_k, float thresh,
                          float *outlier_pc, short *outlier_idx, short *outlier_num, float *inlier_pc, void *stream);
/* get_critical_points + get_critical_pc_non_critical_pc (src/ae_utils.py:12-80) on the device, from (max_val, max_idx) =
 * (np.max, np.argmax)(pre_symmetry, axis=1) as geoadv_ae_critical returns them: (b,c) each.  critical_points (b,c,3) /
 * critical_idx (b,c) int16: the distinct arg-max points of the channels with max_val > 0, the point owning most channels first,
 * zero-padded; critical_num (b) int16; critical_pc (b,n,3): those points padded with the last of them; non_critical_pc (b,n,3):
 * every other point in point order, padded with the last.  Points owning EQUALLY many channels come in descending point
 * index (what a stable sort makes of the reference's np.argsort(counts)[::-1]; numpy's default sort leaves that order to the
 * build).  Any output may be NULL.  c <= 1024, n <= 32768. */
int geoadv_critical_split(int b, int n, int c, const float *pc, const float *max_val, const int *max_idx,
                          float *critical_points, short *critical_idx, short *critical_num, float *critical_pc,
                          float *non_critical_pc, void *stream);

/* ------------------------------------------------------------------------------------------
 * Victim auto-encoder: src/encoders_decoders.py:19-147 with the architecture of
 * src/ae_templates.py:11-39 (5 x [conv1d k=1, BN(inference), ReLU], max over points,
 * FC-ReLU, FC-ReLU, FC).
 * ---------------------------------------------------------------------------------------- */
typedef struct geoadv_ae geoadv_ae;

#define GEOADV_ENC_LAYERS 5
#define GEOADV_DEC_LAYERS 3

/* Host-side description of the weights, in the reference's variable layout
 * (adversary_autoencoder.py:42-51 restores exactly these variables):
 *   enc_w[i]  : [C_i, C_{i+1}]   (tflearn conv_1d W[1,1,Cin,Cout] squeezed), enc_b[i] : [C_{i+1}]
 *   bn_*[i]   : [C_{i+1}]         gamma, beta, moving_mean, moving_variance (eps = 1e-5)
 *   dec_w[k]  : [D_k, D_{k+1}], dec_b[k] : [D_{k+1}]
 * enc_dims = {3, 64, 128, 128, 256, bneck}; dec_dims = {bneck, 256, 256, 3*n_points}.
 * This build supports the template's widths (every C_i, i>=1, a multiple of 32; bneck = 128). */
typedef struct geoadv_ae_weights {
    int n_points;
    int enc_dims[GEOADV_ENC_LAYERS + 1];
    int dec_dims[GEOADV_DEC_LAYERS + 1];
    const float *enc_w[GEOADV_ENC_LAYERS], *enc_b[GEOADV_ENC_LAYERS];
    const float *bn_gamma[GEOADV_ENC_LAYERS], *bn_beta[GEOADV_ENC_LAYERS];
    const float *bn_mean[GEOADV_ENC_LAYERS], *bn_var[GEOADV_ENC_LAYERS];
    const float *dec_w[GEOADV_DEC_LAYERS], *dec_b[GEOADV_DEC_LAYERS];
} geoadv_ae_weights;

/* Uploads (and re-packs for MFMA) the HOST weights.  Allocates device memory; synchronous. */
int  geoadv_ae_create(geoadv_ae **out, const geoadv_ae_weights *host_weights);
void geoadv_ae_destroy(geoadv_ae *ae);

/* Arithmetic of the encoder's four wide layers (the path's dominant kernel; 180 kFLOP per point).
 *   F32    : v_mfma_f32_32x32x2_f32 -- fp32 products, fp32 accumulate (bounded by the 157 TFLOP/s fp32 matrix peak).
 *   BF16X3 : every operand as three bf16 pieces (8 + 8 + 8 bits = the fp32's 24), a product as its six piece products of
 *            weight >= 2^-16, each exact in the fp32 accumulator, on v_mfma_f32_32x32x16_bf16 (csrc/encoder_x3.h).  Error
 *            against float64 = that of an fp32 accumulation in another order (profiles/r05_bf16x3_probe.jsonl); not the
 *            bits of F32.  Non-finite coordinates give NaN activations (inf - inf in the split) where F32 gives inf.
 *   F16X2  : every operand as TWO fp16 pieces (11 + 11 bits; remainder < 2^-23) of power-of-two-scaled values, a product as its
 *            three piece products of weight >= 2^-11, each exact in the fp32 accumulator, on v_mfma_f32_32x32x16_f16: half of
 *            BF16X3's matrix instructions, the same error against float64 (profiles/r06_f16x2_probe.jsonl: 0.44-0.52 units of
 *            2^-24 |a|.|w| rms, the fp32 chain's).  A layer's activations are carried times a power of two s_j -- 2^6 for a
 *            layer whose batch norm has gamma^2 + beta^2 = 1 on average, moved with that magnitude otherwise (what
 *            relu(gamma z + beta) puts out) --, its weights times the power of two that puts the largest in [2^13, 2^14);
 *            RANGE: an activation of 1023.5 x that magnitude or more does not fit (clouds far outside what the victim's
 *            batch norms were made for) -- the kernels check every activation they split, and a cloud that has one gets
 *            +inf latents (so that nothing downstream looks sane) and raises the model's sticky flag: geoadv_ae_status /
 *            geoadv_attack_status then return GEOADV_ERANGE.  Such clouds run under BF16X3.  Not available (set refuses, the
 *            default falls back to BF16X3) for a model with a non-finite weight or a folded constant outside the normal fp32
 *            range.
 * All reproduce themselves bit for bit (forward, recomputing backward, any batch).  The default of handles created from
 * now on (AUTO, the initial setting: F16X2 where available, else BF16X3) / of one handle (set before it is shared with attack
 * handles or threads). */
#define GEOADV_ENC_ARITH_AUTO  (-1)
#define GEOADV_ENC_ARITH_F32    0
#define GEOADV_ENC_ARITH_BF16X3 1
#define GEOADV_ENC_ARITH_F16X2  2
int  geoadv_set_default_encoder_arith(int arith);
int  geoadv_ae_set_encoder_arith(geoadv_ae *ae, int arith);
int  geoadv_ae_encoder_arith(const geoadv_ae *ae);
/* Synchronises `stream`; GEOADV_ERANGE (message in geoadv_last_error; the flag is cleared) if an F16X2 forward of this model since
 * the last call met an activation outside its range, else GEOADV_OK.  Callers that synchronise anyway (to read results) call it. */
int  geoadv_ae_status(const geoadv_ae *ae, void *stream);

/* AdversaryAutoEncoder.reconstruct / AutoEncoder.transform (adversary_autoencoder.py:75-91):
 * pc[b,n,3] -> latent[b,bneck] (may be NULL) and recon[b,n,3] (may be NULL).
 * workspace: geoadv_ae_workspace_bytes(ae,b) bytes of device scratch. */
size_t geoadv_ae_workspace_bytes(const geoadv_ae *ae, int b);
/* Encoder only, with the arg-max of the symmetric pool (what src/ae_utils.py:19-20 derives from
 * get_pre_symmetry_data): latent[b,bneck] = max over points of the last encoder layer, arg_idx[b,bneck] = the
 * LOWEST point index attaining it (np.argmax semantics).  Same workspace as geoadv_ae_forward. */
int geoadv_ae_critical(const geoadv_ae *ae, int b, const float *pc, float *latent, int *arg_idx,
                       void *workspace, void *stream);
int geoadv_ae_forward(const geoadv_ae *ae, int b, const float *pc, float *latent, float *recon,
                      void *workspace, void *stream);
/* AutoEncoder.decode (src/autoencoder.py:191-194; used by interpolate, :178-189): latent[b,bneck] -> recon[b,n,3].
 * Same arithmetic as the decoder half of geoadv_ae_forward: decode(transform(x)) == reconstruct(x) bit for bit.
 * Same workspace as geoadv_ae_forward. */
int geoadv_ae_decode(const geoadv_ae *ae, int b, const float *latent, float *recon, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------
 * The attack loop: AdvAE (src/adv_ae.py:30-251) + Adversary (src/adversary.py:9-57).
 * One handle = one batch slot of `batch` clouds with device-resident state
 * (pert, Adam m/v/beta powers, best-so-far outputs).
 * ---------------------------------------------------------------------------------------- */
typedef struct geoadv_attack geoadv_attack;

#define GEOADV_LOSS_ADV_CHAMFER 0   /* loss_adv_type 'chamfer' (output space), adv_ae.py:88   */
#define GEOADV_LOSS_ADV_LATENT  1   /* loss_adv_type 'latent',  adv_ae.py:85-86,107-116      */
#define GEOADV_LOSS_DIST_CHAMFER 0  /* loss_dist_type 'chamfer', adv_ae.py:98-102            */
#define GEOADV_LOSS_DIST_PERT    1  /* loss_dist_type 'pert',    adv_ae.py:93-97             */

typedef struct geoadv_attack_config {
    int   batch;                    /* conf.batch_size                                        */
    int   loss_adv_type;            /* GEOADV_LOSS_ADV_*                                      */
    int   loss_dist_type;           /* GEOADV_LOSS_DIST_*                                     */
    float max_point_pert_weight;    /* conf.max_point_pert_weight (adv_ae.py:94-95)           */
    float max_point_dist_weight;    /* conf.max_point_dist_weight (adv_ae.py:99-100)          */
    float learning_rate;            /* conf.learning_rate (adv_ae.py:146,152)                 */
    float emd_weight;               /* build-defined: loss_adv += emd_weight*match_cost/N; 0 = off */
    int   all_pairs_source_dist;    /* 0 (default): nn_distance(adv, x) by the exact paired grid search, falling back per
                                     * cloud to the all-pairs kernel, except for tiny batches -- batch * n_points < 10240, i.e.
                                     * up to 4 clouds of 2048 points (GEOADV_SMALL_BATCH_POINTS) -- where the all-pairs kernel alone
                                     * is as fast; 1: always the all-pairs kernel; 2: the grid search at every size.  Same results. */
    int   emd_weight_mode;          /* GEOADV_EMD_FAST (0, default) or GEOADV_EMD_REFERENCE for the EMD term's plan,
                                     * optionally | GEOADV_EMD_DENSE_LEVELS (this handle's EMD sweeps all dense)            */
    /* Alternative code paths with the same results, selected explicitly (never by the environment); all 0 = defaults.
     * The parity tests run every one of them against the default path.                                                */
    int   recompute_backward;       /* 1: the sparse encoder backward re-runs the forward for the critical rows instead
                                     * of reading the ReLU masks the forward kept (the path tied clouds always take)    */
    int   encoder_backward;         /* how the masks are used: GEOADV_ENC_BWD_AUTO (0) = the pool Jacobian (encoder_jac.h) where
                                     * it can be evaluated beside the symmetric Chamfer scan, else the masked backward;
                                     * _MASKED (1) = always back-propagate dz through the critical rows; _JACOBIAN (2) = always
                                     * the Jacobian (a launch of its own where nothing hosts it).  Equal to rounding.       */
    int   separate_adam;            /* 1: the Adam step is its own launch (the path loss_dist_type 'pert' always takes)
                                     * instead of riding in the next forward's point loaders                           */
    int   chamfer_kernel;           /* GEOADV_CHAMFER_AUTO (0: by batch size, GEOADV_SYM_MIN_POINTS), _TWO_SCAN (the public op's kernel),
                                     * _SYMMETRIC (one evaluation per pair serves both directions)                      */
    int   loss_in_scan;             /* 0 (default): the loss / metrics / keep-best and Chamfer-gradient workgroups ride as the last
                                     * workgroups of the symmetric scan's launch where that pays and it can host them (the paired
                                     * search in the launch and two scan workgroups per CU: batch >= 64 at 2048 points; both losses Chamfer, no EMD
                                     * term, batch a multiple of 8, one row super-tile, and a device that deals workgroup i to XCD
                                     * i % 8: checked once per handle) -- they wait for their cloud's workgroups through a counter
                                     * instead of a kernel boundary; 1: always a launch of their own; 2: riding wherever the launch CAN
                                     * host them, paying or not (the parity tests).  Same results, bit for bit. */
} geoadv_attack_config;
#define GEOADV_SMALL_BATCH_POINTS 10240  /* batch * n_points below this: no paired grid search (all_pairs_source_dist = 0)                    */
#define GEOADV_SYM_MIN_POINTS      4096  /* batch * n_points from this on: chamfer_kernel AUTO = the symmetric scan, and with it
                                          * encoder_backward AUTO = the pool Jacobian (it rides in that scan's launch); below: the two-scan
                                          * kernel and the masked backward                                                               */
#define GEOADV_ENC_BWD_AUTO      0
#define GEOADV_ENC_BWD_MASKED    1
#define GEOADV_ENC_BWD_JACOBIAN  2
#define GEOADV_CHAMFER_AUTO      0
#define GEOADV_CHAMFER_TWO_SCAN  1
#define GEOADV_CHAMFER_SYMMETRIC 2

int  geoadv_attack_create(geoadv_attack **out, const geoadv_ae *ae, const geoadv_attack_config *cfg);
void geoadv_attack_destroy(geoadv_attack *at);

/* feed_dict of adv_ae.py:202,213: x = source_pc[B,N,3], gt = target_pc[B,N,3],
 * target_z = target_latent[B,bneck], dist_weight[B].  Device pointers, copied into the handle. */
int geoadv_attack_set_inputs(geoadv_attack *at, const float *source_pc, const float *target_pc,
                             const float *target_latent, const float *dist_weight, void *stream);
/* Adversary.init_pert (adversary.py:27-28): pert <- init[B,N,3] (device).  Also resets the
 * best-so-far bookkeeping of _attack_one_batch (adv_ae.py:197-200).  The Adam slots and beta
 * powers are NOT reset (they are never re-initialised in the reference, adv_ae.py:74) unless
 * reset_optimizer != 0. */
int geoadv_attack_init_pert(geoadv_attack *at, const float *init_pert, int reset_optimizer, void *stream);

/* Runs `iterations` iterations of the hot loop (adv_ae.py:216-246): each is one Adam step on
 * pert followed by the evaluation of the per-cloud metrics of the UPDATED pert; iterations whose
 * 1-based global index is >= thresh take part in the keep-best update (strict '<' on the target
 * reconstruction error).  Everything stays on the device; nothing is synchronised.
 * `first_iteration` is the 0-based index of the first iteration of this call within the current
 * dist-weight run (so a 500-iteration attack may be issued as several calls).
 * metrics_hist: optional device buffer [iterations, 6, B] receiving, per iteration,
 *   loss_adv, loss_dist, loss_pert, loss_max (or max_dist), input_dist, loss_ae  (adv_ae.py:219-221). */
int geoadv_attack_run(geoadv_attack *at, int first_iteration, int iterations, int thresh,
                      float *metrics_hist, void *stream);

/* Results of the keep-best bookkeeping (adv_ae.py:238-249), device -> caller device buffers:
 * metrics[B,5] = loss_adv, loss_dist, source_chamfer_dist, target_nre, target_recon_error
 * (target_nre = target_recon_error / target_ae_loss_ref[b]), adv[B,N,3], recon[B,N,3]. */
int geoadv_attack_get_best(geoadv_attack *at, const float *target_ae_loss_ref,
                           float *metrics, float *adv, float *recon, void *stream);

/* From the next forward on: on != 0 -- nn_distance(adv, x) through the paired grid search (where the handle's size supports it);
 * 0 -- the all-pairs kernel for every cloud.  Same results either way; a caller that sees most clouds handed back
 * (geoadv_attack_search_state: a victim whose perturbations leave the grid cells) switches the search off and saves its cost.
 * Overrides geoadv_attack_config.all_pairs_source_dist, including its small-batch rule. */
int geoadv_attack_set_source_search(geoadv_attack *at, int on);

/* Health of the run since the last set_inputs / init_pert: synchronises the stream and returns GEOADV_EHIP (message in
 * geoadv_last_error) if an in-launch hand-off of the loop ever gave up waiting -- the bounded spin of the dense encoder backward
 * on its cloud's decoder-tail flag; never observed, but a run after it would have used a stale gradient.  geoadv_attack_get_best
 * additionally NaN-fills the metrics of such a run.  Also GEOADV_ERANGE if the victim's F16X2 range guard tripped (geoadv_ae_status).
 * Callers that synchronise anyway (to download results) call this first. */
int geoadv_attack_status(geoadv_attack *at, void *stream);

/* Introspection for tests: copies of the current device state (any pointer may be NULL).
 * pert/adv/recon/grad [B,N,3]; latent [B,bneck]; idx_* [B,N] of the last forward:
 * idx_r1/idx_r2 = nn_distance(recon, gt) indices, idx_a1/idx_a2 = nn_distance(adv, x) indices. */
int geoadv_attack_peek(geoadv_attack *at, float *pert, float *adv, float *recon, float *latent,
                       float *grad, int *idx_r1, int *idx_r2, int *idx_a1, int *idx_a2, void *stream);

/* How nn_distance(adv, x) is being answered: *searched = 1 if the paired grid search is in use for this handle (0: all-pairs
 * kernel, by configuration or batch size), *handed_back = number of clouds of the batch whose pairing the search currently
 * judges too poor (they go through the all-pairs kernel; verdicts of the last forward).  Synchronises the stream. */
int geoadv_attack_search_state(geoadv_attack *at, int *searched, int *handed_back, void *stream);

/* ------------------------------------------------------------------------------------------
 * Victim auto-encoder TRAINING step: PointNetAutoEncoder._create_loss / _setup_optimizer
 * (src/pointnet_ae.py:71-99) driven by AutoEncoder.partial_fit (src/autoencoder.py:105-125) with the
 * architecture of src/ae_templates.py:22-33: encoder BN in TRAINING mode (tflearn batch_normalization:
 * batch statistics over all batch*n_points rows, differentiated through; moving averages updated with
 * `bn_decay`, zero_debias=False), loss = reduce_mean(dist1) + reduce_mean(dist2) of nn_distance(recon, gt) or the approx-EMD
 * match cost (geoadv_train_config.loss),
 * Adam (TF 1.13 ApplyAdam form, beta1 .9, beta2 .999, eps 1e-8) on every trainable variable.
 * One handle = one model replica with a fixed batch size; everything is device resident.
 * ---------------------------------------------------------------------------------------- */
typedef struct geoadv_trainer geoadv_trainer;
typedef struct geoadv_train_config {
    int   batch;            /* conf.batch_size (default_train_params: 50)          */
    float learning_rate;    /* conf.learning_rate (0.0005)                         */
    float bn_decay;         /* encoder b_norm_decay (encoders_decoders.py:20: 0.9) */
    int   loss;             /* conf.loss (pointnet_ae.py:74-79): GEOADV_TRAIN_LOSS_CHAMFER (0) = reduce_mean(dist1) +
                             * reduce_mean(dist2) of nn_distance(recon, gt); GEOADV_TRAIN_LOSS_EMD (1) =
                             * reduce_mean(match_cost(recon, gt, approx_match(recon, gt))), the match held constant in the
                             * backward (approx_match is registered NoGradient, tf_approxmatch.py:19)                  */
    int   max_workgroups;   /* 0 = as many persistent workgroups as the device holds (the layer kernels deal their 64- / 32-row
                             * tiles round-robin over them); > 0 caps them.  Results do not depend on it beyond the order of
                             * the per-workgroup weight-gradient partial sums; the parity tests use small values so that
                             * small shapes run the kernels' multi-tile pipelines                                       */
} geoadv_train_config;
#define GEOADV_TRAIN_LOSS_CHAMFER 0
#define GEOADV_TRAIN_LOSS_EMD     1

/* init: HOST weights (the initial variable values; bn_mean / bn_var = the moving averages).  n_points % 64 == 0. */
int  geoadv_trainer_create(geoadv_trainer **out, const geoadv_ae_weights *init, const geoadv_train_config *cfg);
void geoadv_trainer_destroy(geoadv_trainer *t);
/* partial_fit(X, GT): x, gt device [batch,n,3] (gt NULL = x, the non-denoising case); loss: device float (of the
 * PRE-update weights, like the fetched `self.loss`), recon: device [batch,n,3] or NULL. */
int geoadv_trainer_step(geoadv_trainer *t, const float *x, const float *gt, float *loss, float *recon, void *stream);
/* The two halves of a step, for data-parallel training: forward_backward leaves d loss / d variable in the flat
 * gradient buffer; the caller sum-all-reduces that buffer over the ranks (RCCL) and calls apply with
 * grad_scale = 1 / world_size. */
int geoadv_trainer_forward_backward(geoadv_trainer *t, const float *x, const float *gt, float *loss, float *recon, void *stream);
int geoadv_trainer_apply(geoadv_trainer *t, float grad_scale, void *stream);
/* Synchronised batch norm for data-parallel training (so that `world` ranks with `batch` clouds each take EXACTLY the
 * step of one replica with world * batch clouds): the encoder's only coupling between rows is a pair of per-channel
 * sums per layer and direction, so the step is cut into geoadv_trainer_num_phases() phases; after phase p the caller
 * sum-all-reduces the *count doubles geoadv_trainer_exchange(t, p, ...) points at (RCCL; <= 4 KB, latency-bound) and
 * runs phase p + 1.  After the last phase: all-reduce the flat gradient buffer, geoadv_trainer_apply(t, 1.0f), and add
 * up the ranks' geoadv_trainer_fetch losses.  geoadv_trainer_set_world(t, world) switches the mode (1 = off). */
int geoadv_trainer_set_world(geoadv_trainer *t, int world);
int geoadv_trainer_num_phases(void);
int geoadv_trainer_run_phase(geoadv_trainer *t, int phase, const float *x, const float *gt, void *stream);
int geoadv_trainer_exchange(geoadv_trainer *t, int phase, double **buf, size_t *count);
int geoadv_trainer_fetch(geoadv_trainer *t, float *loss, float *recon, void *stream);
/* Device pointers of the flat parameter / gradient buffers (`count` floats each) and where each variable sits:
 * offsets26 = enc_w[5], enc_b[5], bn_gamma[5], bn_beta[5], dec_w[3], dec_b[3] (in floats). */
int geoadv_trainer_buffers(geoadv_trainer *t, float **params, float **grads, size_t *count);
int geoadv_trainer_layout(const geoadv_trainer *t, size_t *offsets26);
/* Downloads the current variables (and moving averages) into the HOST buffers `dst` points to -- what
 * saver.save writes (autoencoder.py:213-215); feed them to geoadv_ae_create to attack the trained model. */
int geoadv_trainer_export(geoadv_trainer *t, const geoadv_ae_weights *dst, void *stream);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * enable is a bit mask of GEOADV_PROF_* classes (bit k = class k, -1 = all, 0 = off; enabling resets
 * the totals): every geoadv_attack_run iteration brackets the selected kernels with events (a fixed
 * pool is recycled; when it runs dry the stream is synchronised once).  geoadv_attack_profile_read synchronises the stream and returns, for
 * kernel class `which` (GEOADV_PROF_*), the number of launches timed and their total ms. */
#define GEOADV_PROF_ENCODER_FWD 0
#define GEOADV_PROF_DECODER_FWD 1
#define GEOADV_PROF_CHAMFER_FWD 2
#define GEOADV_PROF_LOSS_GRAD   3
#define GEOADV_PROF_DECODER_BWD 4
#define GEOADV_PROF_ENCODER_BWD 5
#define GEOADV_PROF_ADAM        6
#define GEOADV_PROF_COUNT       7
int geoadv_attack_profile(geoadv_attack *at, int enable);
int geoadv_attack_profile_read(geoadv_attack *at, int which, int *launches, float *total_ms);
/* roctx ranges "geoadv:<class>" around the launches of every kernel class (for rocprofv3 --marker-trace timelines);
 * libroctx64 is looked up at run time.  Returns GEOADV_EINVAL if it cannot be found. */
int geoadv_attack_markers(geoadv_attack *at, int enable);
/* Time only every stride-th launch of each selected class (a pair of events between two dependent kernels costs ~1 us of
 * GPU time; sampling keeps a timed region honest).  Default 1. */
int geoadv_attack_profile_stride(geoadv_attack *at, int stride);

#ifdef __cplusplus
}
#endif
#endif /* GEOADV_H */
