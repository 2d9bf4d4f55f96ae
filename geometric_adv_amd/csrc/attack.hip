// The attack loop of AdvAE (src/adv_ae.py:191-251) + Adversary (src/adversary.py) on one GPU:
// device-resident state, one forward per iteration, no host synchronisation inside the loop.
//
// Reference schedule per iteration (adv_ae.py:216-246): sess.run(attack_op) = forward(pert) ->
// backward -> Adam, then a SECOND full forward to evaluate the metrics of the updated pert (and a
// third/fourth run for adv / reconstruct once iteration+1 >= thresh).  Here an iteration is
//   backward (from the cached forward) -> Adam -> forward(pert_new) -> metrics + keep-best,
// i.e. exactly one forward per iteration (plus one after init_pert); the observable outputs are
// index-aligned with the reference (metrics of iteration k describe the state after k+1 updates).
#include "ae.h"
#include "chamfer_sym.h"
#include "chamfer_grad.h"
#include "chamfer_grid.h"
#include "encoder_jac.h"
#include "decoder_tail.h"
#define LC_STAMP(K, I) GA_STAMP(K, I)
#include "loss_cgrad.h"
#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <vector>

#pragma clang fp contract(off)

namespace geoadv {

// ---- from the other translation units ------------------------------------------------------
struct ChamferScan { const float *query; const float *target; float *dist; int *idx; int nq, nt; const int *need; };
int launch_chamfer_scans(const ChamferScan *scans, int nscan, int b, hipStream_t stream);
int launch_encoder_jac(const DeviceAE &A, int b, const JacArgs &a, hipStream_t stream);
int launch_encoder_bwd_dense(const DeviceAE &A, int b, const float *adv, const float *z, const int *zcnt, const float *dz,
                             const int *dense_flag, float *g_enc, hipStream_t stream);
int launch_decoder_fc2_bwd(const DeviceAE &A, int b, const float *g_recon, float *partial, hipStream_t stream);
int launch_decoder_tail_dense(const DeviceAE &A, const TailDenseArgs &a, hipStream_t stream);
bool chamfer_grid_rides(int n);
int launch_chamfer_grid(const float *P, const float *Q, float *d1, int *i1, float *d2, int *i2, int b, int n, int *need, int call, const float *box,
                        hipStream_t stream);
bool chamfer_grid_supports(int n, int m);
int launch_latent_decode_and_grid(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z, int *crit,
                                  int *zcnt, int *dense, float *d1, float *d2, const float *P, const float *Q, float *gd1, int *gi1,
                                  float *gd2, int *gi2, int n, int *need, int call, const float *box, hipStream_t stream);
int launch_chamfer_grid_box(const float *Q, int b, int n, float *box, hipStream_t stream);
struct ForwardScratch {
    float *pmax; int *parg; int *pcnt; float *z; int *crit; int *zcnt; int *dense; float *d1, *d2; size_t bytes;
};
ForwardScratch carve_forward_scratch(void *base, int b, int n_points);
int launch_encoder_fwd(const DeviceAE &A, int b, const float *x, const float *pert, float *adv_out, float *pmax,
                       int *parg, int *pcnt, unsigned *masks, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                       const FusedAdam *fused = nullptr);
int encoder_mask_words();
int launch_latent_decode(const DeviceAE &A, int b, const float *pmax, const int *parg, const int *pcnt, float *z,
                         int *crit, int *zcnt, int *dense, float *d1, float *d2, hipStream_t stream);
int launch_decoder_fc2(const DeviceAE &A, int b, const float *d2, float *recon, hipStream_t stream, unsigned long long *fill = nullptr,
                       size_t fill_count = 0);
int launch_decoder_bwd(const DeviceAE &A, int b, const float *g_recon, const float *d1, const float *d2, float *partial,
                       float *dz, hipStream_t stream, const int *crit = nullptr, const float *jac = nullptr, const int *dense = nullptr,
                       float *g_enc = nullptr);
int decoder_bwd_chunks(const DeviceAE &A);
int launch_encoder_bwd(const DeviceAE &A, int b, const float *adv, const int *crit_rows, const float *z,
                       const int *zcnt, const float *dz, const int *dense_flag, float *g_enc, const unsigned *masks,
                       hipStream_t stream);

__global__ __launch_bounds__(256) void loss_metrics_kernel(LossArgs a) { loss_metrics_body(a, blockIdx.x, gridDim.x); }

// chamfer_dist = reduce_mean(dist1, axis=1) + reduce_mean(dist2, axis=1) (get_dists_per_point.py:75, prepare_indices_for_attack.py:114)
// as an operator, in EXACTLY the summation order of loss_metrics_body (thread t adds elements t, t+256, ... in ascending
// order; xor-shuffle tree; waves 0..3 left to right; sum * (1/n) per direction): the Chamfer distance recomputed from a saved
// adversarial cloud equals the loop's own source_chamfer_dist metric bit for bit -- the reference's sanity check
// (get_dists_per_point.py:114-115) compares them with np.array_equal.
__global__ __launch_bounds__(256) void chamfer_per_pc_kernel(int n, int m, const float *d1, const float *d2, float *out) {
    __shared__ float shf[4][8];
    __shared__ int shi[4][2];
    const int b = blockIdx.x, t = threadIdx.x;
    CloudRed r;
    r.s1 = r.s2 = r.s3 = r.s4 = r.sp = 0.f;
    r.ma = r.mp = -1.f;
    r.ja = r.jp = INT_MAX;
    for (int j = t; j < n; j += 256) r.s3 += d1[(size_t)b * n + j];
    for (int j = t; j < m; j += 256) r.s4 += d2[(size_t)b * m + j];
    r = block_reduce(r, shf, shi);
    if (t == 0) out[b] = r.s3 * (1.0f / (float)n) + r.s4 * (1.0f / (float)m);
}

// ------------------------------------------------------------------------------------------
// Chamfer gradient w.r.t. the FIRST cloud only, with per-cloud constant upstream gradients
// (mean over points => 1/n, times dist_weight for the source-distance term) -- the two uses the
// attack has (adv_ae.py:120-121,131-133 differentiated by adv_ae.py:153).  Same CPU-ordered
// accumulation as chamfer.hip (tf_nndistance.cpp:130-163): own term, then scatter terms in
// ascending index.  grid = (clouds, problems).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CGA_THREADS) void chamfer_grad_attack_kernel(CGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const CGradProblem pr = a.pr[blockIdx.y];
    const int b = blockIdx.x, n = a.n;
    const float wb = pr.w ? pr.w[b] : 1.0f;
    const float gd = wb * (1.0f / (float)n);          // d(mean over n points)/d dist, times dist_weight
    GradSide s;
    s.n_own = n; s.n_oth = n;
    s.own = pr.p + (size_t)b * n * 3; s.oth = pr.q + (size_t)b * n * 3;
    s.match_own = pr.idx1 + (size_t)b * n; s.match_oth = pr.idx2 + (size_t)b * n;
    s.gd_own = nullptr; s.gd_oth = nullptr; s.gd_own_s = gd; s.gd_oth_s = gd;
    s.jstar = (pr.jstar && pr.extra_w > 0.f) ? pr.jstar[b] : -1;
    s.extra = wb * pr.extra_w;                        // max_point_dist_weight * max_j dist1[j] (adv_ae.py:100)
    s.gout = pr.g + (size_t)b * n * 3;
    chamfer_grad_side<true, CGA_THREADS>(s, lds, a.P);
}

// grid = (clouds, problems * H)
__global__ __launch_bounds__(CGA_THREADS) void chamfer_grad_attack_fx_kernel(CGradArgs a, int H) {
    extern __shared__ __attribute__((aligned(16))) unsigned fx_lds[];
    cgrad_fx_body(a, blockIdx.y / H, blockIdx.x, blockIdx.y % H, H, fx_lds);
}

// The per-cloud losses and the Chamfer gradients read the same NN results and do not depend on each other (unless the
// max-distance term is on: it needs the arg-max the loss pass finds), so one launch does both: grid = (clouds,
// 1 + problems * H); row 0 = losses / metrics / keep-best (its four upper waves leave at once), rows 1.. = gradients.
__global__ __launch_bounds__(CGA_THREADS) void loss_cgrad_kernel(LossArgs la, CGradArgs ca, int H) {
    extern __shared__ __attribute__((aligned(16))) unsigned lc_lds[];
    GA_STAMP(0, 0);
    loss_cgrad_block(la, ca, H, blockIdx.x, gridDim.x, blockIdx.y, lc_lds);
    GA_STAMP(0, 7);
}

// ------------------------------------------------------------------------------------------
// Adam on pert (tf.train.AdamOptimizer defaults, TF 1.13 ApplyAdam form; adv_ae.py:152-153).
// g = g_enc + g_dist, g_dist either the Chamfer source-distance gradient (buffer) or the
// perturbation-norm gradient computed here.
// ------------------------------------------------------------------------------------------
struct AdamArgs {
    int n, B;
    float *pert, *m, *v;
    float *g_enc;                    // read, then zeroed for the next iteration's sparse scatter
    const float *g_dist;             // null in 'pert' mode
    const float *w, *losses;         // [B], [8][B]
    const int *jstar;                // [2][B]
    int loss_dist_type;
    float mp_pert_w;
    float alpha, one_minus_b1, one_minus_b2, eps;
    float *grad_out;                 // optional copy of g (tests)
    const float *x;                  // source clouds
    float *adv_out;                  // adv = x + pert_new, ready for the next forward (Adversary.attack, adversary.py:35)
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const size_t per = (size_t)a.n * 3;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per * a.B) return;
    const int b = (int)(e / per);
    float g = a.g_enc[e];
    a.g_enc[e] = 0.f;
    if (a.loss_dist_type == GEOADV_LOSS_DIST_PERT) {
        const float p = a.pert[e];
        const float wb = a.w[b];
        float gp = wb * (p / a.losses[2 * a.B + b]);                   // d sqrt(sum pert^2) = pert / norm
        if (a.mp_pert_w > 0.f) {
            const int pt = (int)((e % per) / 3);
            if (pt == a.jstar[a.B + b]) gp += wb * a.mp_pert_w * (p / a.losses[6 * a.B + b]);
        }
        g += gp;
    } else {
        g += a.g_dist[e];
    }
    if (a.grad_out) a.grad_out[e] = g;
    float m = a.m[e], v = a.v[e];
    m += (g - m) * a.one_minus_b1;
    v += (g * g - v) * a.one_minus_b2;
    a.m[e] = m; a.v[e] = v;
    const float pnew = a.pert[e] - (m * a.alpha) / (sqrtf(v) + a.eps);
    a.pert[e] = pnew;
    a.adv_out[e] = a.x[e] + pnew;
}

__global__ void axpy_kernel(float *y, const float *x, float a, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) y[i] = fmaf(a, x[i], y[i]);
}

__global__ void fill_kernel(float *p, float v, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// metrics[B][5] = loss_adv, loss_dist, source_chamfer_dist, target_nre, target_recon_error
// `failed`: the spin-timeout word of the merged tail + dense launch (never set in any run so far): a result computed after a
// hand-off gave up must not look like a result
__global__ void best_metrics_kernel(int B, const float *best_metrics, const float *best_err, const float *ref, float *out, const int *failed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (failed && *failed) {
        for (int k = 0; k < 5; ++k) out[b * 5 + k] = NAN;
        return;
    }
    out[b * 5 + 0] = best_metrics[b * 4 + 0];
    out[b * 5 + 1] = best_metrics[b * 4 + 1];
    out[b * 5 + 2] = best_metrics[b * 4 + 2];
    out[b * 5 + 3] = best_metrics[b * 4 + 3] / ref[b];          // adv_ae.py:241
    out[b * 5 + 4] = best_err[b];                                // adv_ae.py:249
}

// Workgroup i of a launch runs on XCD i % 8 (each XCD has its own L2): what the loss riders' same-XCD hand-off relies on.
// Recorded, not assumed: 64 workgroups store the XCC id they ran on.
__global__ void xcd_probe_kernel(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = v & 0xf;
    }
}

}  // namespace geoadv

using namespace geoadv;

struct geoadv_attack {
    const geoadv_ae *ae;
    geoadv_attack_config cfg;
    int B, n;
    void *arena; size_t arena_bytes;
    // device state
    float *x, *gt, *tz, *w;
    float *pert, *m, *v;
    float *adv, *recon, *g_recon, *g_dist, *g_enc, *grad_last;
    float *r1, *r2, *a1, *a2; int *ir1, *ir2, *ia1, *ia2;
    ForwardScratch fs;
    float *dz, *dec_partial;
    float *losses; int *jstar;
    float *best_err, *best_metrics, *best_adv, *best_recon;
    float *emd_temp, *emd_cost, *emd_g1;   // only when cfg.emd_weight > 0
    float *sym_ws;                   // row / column partials of the symmetric Chamfer kernel
    unsigned long long *row64;       // [2][B][n] packed row minima of the symmetric scan's atomic form (small batches, chamfer_sym.h)
    bool row64_filled;               // ... set to all ones by this forward's FC2 launch
    bool cgrad_done;                 // the cached forward's loss launch also produced the Chamfer gradients
    bool counted = false;            // this handle is counted in ae->attack_refs (geoadv_ae_set_encoder_arith refuses a switch under it)
    bool chamfer_prune;              // nn_distance(adv, x) through the paired grid search (cfg.all_pairs_source_dist = 0, the default)
    int *need_adv[2];                // [8 B] each: clouds the grid search handed back to the all-pairs kernel.  When the search
                                     // shares the all-pairs launch, call k reads [k & 1] (the verdicts of call k - 1) and
                                     // writes [(k + 1) & 1]; a search in its own launch (n > 4096) uses [0] in place
    float *x_box;                    // [B][6] bounding boxes of the source clouds (the grid of the paired search)
    int grid_calls;                  // running number of grid-search launches (paces the retries of clouds that gave up)
    unsigned *masks;                 // [B][n][mask words] ReLU masks of the cached forward, or null (backward recomputes)
    float *jac;                      // [B][128][3] pool Jacobian of the cached forward (encoder_jac.h), or null: the output-space
                                     // attack's encoder backward is then 128 x 3 multiply-adds in the decoder backward's tail
    bool jac_valid;                  // ... computed for the cached forward
    unsigned *tail_ready;            // [B] + 1 word: hand-off flags of the merged tail + dense launch, the spin-timeout word
    unsigned tail_epoch;             // (host) steps launched so far: the flag value of the current one
    unsigned *loss_done;             // [B] arrival counters of the loss riders in the symmetric scan's launch (loss_cgrad.h), never reset
    unsigned loss_target;            // (host) arrivals a cloud's counter has seen after the launches so far
    bool host_loss_force;            // cfg.loss_in_scan == 2: wherever the launch can host them, not only where it pays
    bool host_loss;                  // loss + gradient workgroups ride in the scan's launch (cfg.loss_in_scan, and the device deals
                                     // workgroup i to XCD i % 8: checked once at creation)
    int cus;                         // compute units of THIS device (hipDeviceAttributeMultiprocessorCount at create): the merged tail +
                                     // dense launch spins on in-launch flags and needs all its workgroups resident at once
    bool chamfer_sym;
    // host state
    float beta1_pow, beta2_pow;
    bool fwd_valid;
    bool adv_valid;                  // adv == x + pert already (written by the Adam kernel)
    bool fuse_adam;                  // the Adam step rides in the next forward's point loads (cfg.separate_adam: own launch)
    bool adam_pending;               // ... and one is waiting there
    FusedAdam pending;
    // profiling
    unsigned prof_mask;
    int prof_stride;                 // time every prof_stride-th launch of a selected class (1 = every launch)
    unsigned prof_seen[GEOADV_PROF_COUNT];
    std::vector<hipEvent_t> ev;      // pool
    int ev_used;
    struct Mark { int which, e0, e1; };
    std::vector<Mark> marks;
    double prof_ms[GEOADV_PROF_COUNT];
    int prof_n[GEOADV_PROF_COUNT];
    hipStream_t prof_stream;
    bool markers;                    // roctx ranges around every kernel class (geoadv_attack_markers)
};

namespace {

inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

int prof_flush(geoadv_attack *at) {
    if (at->marks.empty()) return GEOADV_OK;
    GA_HIP(hipStreamSynchronize(at->prof_stream));
    for (const auto &mk : at->marks) {
        float ms = 0;
        GA_HIP(hipEventElapsedTime(&ms, at->ev[mk.e0], at->ev[mk.e1]));
        at->prof_ms[mk.which] += ms;
        at->prof_n[mk.which] += 1;
    }
    at->marks.clear();
    at->ev_used = 0;
    return GEOADV_OK;
}

// roctx ranges (tracing): libroctx64 is looked up at run time -- the library has no link-time dependency on the
// tracer -- and the ranges show up in `rocprofv3 --marker-trace` / rocprof-compute timelines as "geoadv:<class>".
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void *h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
const char *const kClassNames[GEOADV_PROF_COUNT] = {"geoadv:encoder_fwd", "geoadv:decoder_fwd", "geoadv:chamfer_fwd", "geoadv:loss_grad",
                                                    "geoadv:decoder_bwd", "geoadv:encoder_bwd", "geoadv:adam"};

// Two ways to time a class.  Bracketing (default): hipEventRecord before and after the scope's launches -- the interval
// includes the dispatch gaps around them (~2 us each between dependent kernels).  Kernel-timed (`kernel` = true, one launch
// per scope): the scope only reserves the two events and the launcher hands them to hipExtLaunchKernel, which stamps the
// kernel's own begin / end -- the quantity rocprofv3 --kernel-trace reports, so the two agree.
struct ProfScope {
    geoadv_attack *at; int which; int e0, e1; hipStream_t st; bool on, kernel, marked;
    ProfScope(geoadv_attack *a, int w, hipStream_t s, bool kernel_timed = false)
        : at(a), which(w), e0(-1), e1(-1), st(s), on((a->prof_mask >> w) & 1u), kernel(kernel_timed),
          marked(a->markers && roctx().push != nullptr) {
        if (marked) roctx().push(kClassNames[w]);
        if (on && at->prof_stride > 1) on = (at->prof_seen[w]++ % (unsigned)at->prof_stride) == 0;
        if (!on) return;
        if (at->ev_used + 2 > (int)at->ev.size()) prof_flush(at);
        e0 = at->ev_used++;
        if (kernel) e1 = at->ev_used++;
        else (void)hipEventRecord(at->ev[e0], st);
    }
    hipEvent_t start() const { return on && kernel ? at->ev[e0] : nullptr; }
    hipEvent_t stop() const { return on && kernel ? at->ev[e1] : nullptr; }
    ~ProfScope() {
        if (marked) roctx().pop();
        if (!on) return;
        if (!kernel) {
            e1 = at->ev_used++;
            (void)hipEventRecord(at->ev[e1], st);
        }
        at->marks.push_back({which, e0, e1});
    }
};


// forward(pert): encoder -> latent/decoder -> both Chamfer problems -> per-cloud losses (+ metrics / keep-best).
int do_forward(geoadv_attack *at, float *hist_slot, int keep, hipStream_t st) {
    const DeviceAE &A = at->ae->d;
    const int B = at->B, n = at->n;
    const ChamferScan sc_recon[2] = {{at->recon, at->gt, at->r1, at->ir1, n, n}, {at->gt, at->recon, at->r2, at->ir2, n, n}};
    const ChamferScan sc_adv[2] = {{at->adv, at->x, at->a1, at->ia1, n, n}, {at->x, at->adv, at->a2, at->ia2, n, n}};
    // nn_distance(adv, x): adv = x + pert and most points barely move, so the exact grid search seeded with the pairing
    // (chamfer_grid.hip) answers it for a fraction of the all-pairs cost; clouds whose pairing has become poor raise
    // their `need` flag and are redone by the all-pairs launch below (same results either way)
    const bool pruned = at->chamfer_prune && chamfer_grid_supports(n, n);
    const bool adv_chamfer = at->cfg.loss_adv_type == GEOADV_LOSS_ADV_CHAMFER;
    const bool dist_chamfer = at->cfg.loss_dist_type == GEOADV_LOSS_DIST_CHAMFER;
    const bool max_term = dist_chamfer && at->cfg.max_point_dist_weight > 0.f;   // the gradient needs the loss pass's arg-max first
    const bool loss_fused = (adv_chamfer || dist_chamfer) && !max_term && n <= CG_FX_MAX_N_PLANE;   // loss_cgrad_kernel below
    const bool merge_in_loss = loss_fused && adv_chamfer && dist_chamfer;
    // narrow column slices (small batches): row minima folded into packed words by atomics; the FC2 launch fills them on its way
    const bool use_row64 = merge_in_loss && at->chamfer_sym && at->row64 != nullptr && chamfer_sym_packs_rows((long)B * (pruned ? 1 : 2), n, n);
    SymPartials part{nullptr, nullptr, 1, B, false, use_row64 ? at->row64 : nullptr};
    {
        ProfScope ps(at, GEOADV_PROF_ENCODER_FWD, st, true);
        if (int rc = launch_encoder_fwd(A, B, at->x, at->pert, (at->adv_valid && !at->adam_pending) ? nullptr : at->adv, at->fs.pmax,
                                        at->fs.parg, at->fs.pcnt, at->masks, st, ps.start(), ps.stop(),
                                        at->adam_pending ? &at->pending : nullptr)) return rc;
        at->adam_pending = false;
        at->adv_valid = true;
    }
    // The paired search needs nothing the network produces and ~11-16 us per workgroup (latency-bound: counting sort, cell
    // walks).  With the symmetric scan (large batches) it shares the launch of the all-pairs scan of (recon, target), which is
    // longer and which it cannot slow down by much -- in the latent_decode launch, where it rode until round 2, it was the
    // longer half (in-kernel stamps: 16.1 against 11.5 us at B = 32).  The scans of (adv, source) in that same launch therefore
    // follow the verdicts of the PREVIOUS call (GridArgs::need_prev).  Small batches (two-scan kernel: 16-wave workgroups that
    // fill a CU's registers, so the search's workgroups would only start when the scans are done) keep it beside
    // latent_decode (a second stream for it, forked after the encoder and joined before the scans, was measured: + 25 us per
    // iteration at B = 1 ... 16 -- each cross-stream event costs more than the whole search); clouds of more than GR_MAX_N
    // points keep the search in a launch of its own, ahead of the scans.
    const bool rides_scan = pruned && chamfer_grid_rides(n) && at->chamfer_sym;
    const bool rides_latent = pruned && chamfer_grid_rides(n) && !at->chamfer_sym;
    const int call = at->grid_calls;
    int *need_new = at->need_adv[rides_scan ? (call + 1) & 1 : 0];
    const int *need_scan = at->need_adv[rides_scan ? call & 1 : 0];      // what the all-pairs kernels act on in this call
    if (pruned) at->grid_calls++;
    {
        ProfScope ps(at, GEOADV_PROF_DECODER_FWD, st);
        if (pruned && !rides_scan && !rides_latent) {   // large clouds: the search takes a CU's whole LDS, so it gets its own launch
            if (int rc = launch_chamfer_grid(at->adv, at->x, at->a1, at->ia1, at->a2, at->ia2, B, n, need_new, call, at->x_box, st)) return rc;
        }
        if (rides_latent) {
            if (int rc = launch_latent_decode_and_grid(A, B, at->fs.pmax, at->fs.parg, at->fs.pcnt, at->fs.z, at->fs.crit, at->fs.zcnt,
                                                       at->fs.dense, at->fs.d1, at->fs.d2, at->adv, at->x, at->a1, at->ia1, at->a2,
                                                       at->ia2, n, need_new, call, at->x_box, st)) return rc;
        } else if (int rc = launch_latent_decode(A, B, at->fs.pmax, at->fs.parg, at->fs.pcnt, at->fs.z, at->fs.crit, at->fs.zcnt,
                                                 at->fs.dense, at->fs.d1, at->fs.d2, st)) return rc;
        if (int rc = launch_decoder_fc2(A, B, at->fs.d2, at->recon, st, use_row64 ? at->row64 : nullptr, use_row64 ? 2 * (size_t)B * n : 0)) return rc;
    }
    // The pool Jacobian of THIS forward (the next step's encoder backward, encoder_jac.h) beside the symmetric scan.  Where the
    // two-scan kernel runs instead (small batches) there is no launch long enough to hide it in, and as a launch of its own
    // (8.2 us + the 3.6 us look for tied clouds + two boundaries) it costs what the masked backward costs (11.6 us + one): those
    // batches keep the masked backward.
    const JacArgs jargs{n, at->masks, at->fs.crit, at->fs.z, at->fs.dense, at->jac};
    const bool jac_rides = at->jac && at->chamfer_sym;
    at->jac_valid = jac_rides;
    if (at->jac && !jac_rides && at->cfg.encoder_backward == GEOADV_ENC_BWD_JACOBIAN) {   // forced: a launch of its own
        ProfScope ps(at, GEOADV_PROF_ENCODER_BWD, st);
        if (int rc = launch_encoder_jac(A, B, jargs, st)) return rc;
        at->jac_valid = true;
    }
    // arguments of the loss / metrics / keep-best pass and of the two Chamfer gradients, for row minima that arrive as `p` says
    auto fill_loss = [&](const SymPartials &p, LossArgs &la, CGradArgs &ca, int &np) {
        la.n = n; la.loss_adv_type = at->cfg.loss_adv_type; la.loss_dist_type = at->cfg.loss_dist_type;
        la.mp_pert_w = at->cfg.max_point_pert_weight; la.mp_dist_w = at->cfg.max_point_dist_weight;
        la.r1 = at->r1; la.r2 = at->r2; la.a1 = at->a1; la.a2 = at->a2; la.pert = at->pert;
        la.z = at->fs.z; la.tz = at->tz; la.w = at->w; la.losses = at->losses; la.jstar = at->jstar;
        la.emd_cost = at->emd_temp ? at->emd_cost : nullptr; la.emd_weight = at->cfg.emd_weight;
        la.dz_latent = at->dz; la.hist = hist_slot; la.keep = keep; la.best_err = at->best_err;
        la.best_metrics = at->best_metrics; la.adv = at->adv; la.recon = at->recon;
        la.best_adv = at->best_adv; la.best_recon = at->best_recon;
        la.part = p; la.a1_need = pruned ? need_scan : nullptr; la.a1_all = pruned ? 0 : 1; la.r1_out = at->r1; la.a1_out = at->a1;
        // the Chamfer gradients the next step starts with ride in the same launch (see loss_cgrad_kernel)
        np = 0;
        if (adv_chamfer) {
            ca.pr[np] = CGradProblem{at->recon, at->gt, at->ir1, at->ir2, at->g_recon, nullptr, nullptr, 0.f};
            if (p.deferred) {
                ca.pr[np].part_d = p.rowpart_d; ca.pr[np].part_i = p.rowpart_i; ca.pr[np].part_slices = p.slices;
                ca.pr[np].part_need = nullptr; ca.pr[np].idx1_out = at->ir1; ca.pr[np].part_w = p.row64;
            }
            ++np;
        }
        if (dist_chamfer) {
            ca.pr[np] = CGradProblem{at->adv, at->x, at->ia1, at->ia2, at->g_dist, at->w, at->jstar, 0.f};
            if (p.deferred) {                             // pair 1's partials follow pair 0's B clouds
                const size_t off = (size_t)B * p.slices * n;
                ca.pr[np].part_d = p.rowpart_d + off; ca.pr[np].part_i = p.rowpart_i + off; ca.pr[np].part_slices = p.slices;
                ca.pr[np].part_need = pruned ? need_scan : nullptr; ca.pr[np].idx1_out = at->ia1;
                ca.pr[np].part_w = p.row64 ? p.row64 + (size_t)B * n : nullptr;
            }
            ++np;
        }
        ca.n = n; ca.P = 0;
    };
    struct LossCtx { void (*fill)(void *, const SymPartials &, LossArgs &, CGradArgs &, int &); void *self; };
    LossCtx lctx{[](void *f, const SymPartials &p, LossArgs &la, CGradArgs &ca, int &np) { (*static_cast<decltype(fill_loss) *>(f))(p, la, ca, np); }, &fill_loss};
    LossRider lr;
    lr.blocks = 0; lr.patch = nullptr; lr.ctx = nullptr; lr.force = false;
    bool want_host = false, hosted = false;
    {
        ProfScope ps(at, GEOADV_PROF_CHAMFER_FWD, st);
        if (at->chamfer_sym) {   // one distance evaluation per pair serves both directions
            const ChamferPair pairs[2] = {{at->recon, at->gt, at->r1, at->ir1, at->r2, at->ir2},
                                          {at->adv, at->x, at->a1, at->ia1, at->a2, at->ia2}};
            const GridArgs rider{at->adv, at->x, at->a1, at->ia1, at->a2, at->ia2, n, need_new, need_scan, call, at->x_box};
            JacRider jr;
            jr.j = jargs; jr.A = A; jr.first_block = 0; jr.blocks = 0; jr.raise_prio = 0;
            // The loss + gradient workgroups as the last riders of this launch (loss_cgrad.h) where it can host them: they wait for
            // their cloud's workgroups through a counter instead of a kernel boundary.  (Not with the EMD term: its launch lies between.)
            if (at->host_loss && merge_in_loss && !at->emd_temp) {
                lr.H = cgrad_fx_parts(n); lr.rows = 1 + 2 * lr.H; lr.done = at->loss_done; lr.target = at->loss_target;
                lr.spin_timeout = reinterpret_cast<int *>(at->tail_ready + B);
                lr.ctx = &lctx; lr.force = at->host_loss_force;
                lr.patch = [](LossRider &r, const SymPartials &p, void *c) {
                    LossCtx &x = *static_cast<LossCtx *>(c);
                    int np_ = 0;
                    x.fill(x.self, p, r.la, r.ca, np_);
                };
                want_host = true;
            }
            // the row minima leave the scan as one (distance, index) partial per column slice; when the loss launch below is the
            // fused one with both Chamfer gradients inside, it merges them on its way in (no second Chamfer launch)
            if (int rc = launch_chamfer_sym_loop(pairs, 2, B, n, n, at->sym_ws, pruned ? need_scan : nullptr, rides_scan ? &rider : nullptr,
                                                 jac_rides ? &jr : nullptr, merge_in_loss ? &part : nullptr, st, want_host ? &lr : nullptr)) return rc;
            hosted = want_host && lr.blocks > 0;
            if (hosted) { at->loss_target = lr.target; at->cgrad_done = true; }
        } else {
            ChamferScan all[4] = {sc_recon[0], sc_recon[1], sc_adv[0], sc_adv[1]};
            if (pruned) all[2].need = all[3].need = need_scan;         // only the clouds the grid search handed back
            if (int rc = launch_chamfer_scans(all, 4, B, st)) return rc;
        }
    }
    if (at->emd_temp) {   // approx_match is NoGradient (tf_approxmatch.py:19): the plan is recomputed every forward; the loop
        ProfScope ps(at, GEOADV_PROF_CHAMFER_FWD, st);   // needs only its cost and d cost / d recon, so the plan itself is never stored
        if (int rc = geoadv_emd_cost_grad1_mode(at->cfg.emd_weight_mode, B, n, n, at->recon, at->gt, at->emd_cost, at->emd_g1, at->emd_temp, st)) return rc;
    }
    if (!hosted) {
        ProfScope ps(at, GEOADV_PROF_LOSS_GRAD, st);
        LossArgs la;
        CGradArgs ca;
        int np = 0;
        fill_loss(part, la, ca, np);
        at->cgrad_done = false;
        if (loss_fused) {
            const int H = cgrad_fx_parts(n);
            loss_cgrad_kernel<<<dim3(B, 1 + np * H), CGA_THREADS, loss_cgrad_lds_bytes(n), st>>>(la, ca, H);
            at->cgrad_done = true;
        } else {
            loss_metrics_kernel<<<B, 256, 0, st>>>(la);
        }
        GA_LAUNCH_CHECK();
    }
    at->fwd_valid = true;
    return GEOADV_OK;
}

int pow2_ge(int v) { int p = 2; while (p < v) p <<= 1; return p; }

int launch_cgrad(const CGradProblem *pr, int np, int B, int n, hipStream_t st) {
    CGradArgs ca;
    for (int i = 0; i < np; ++i) ca.pr[i] = pr[i];
    ca.n = n; ca.P = pow2_ge(n);
    if (n <= CG_FX_MAX_N_PLANE) {
        const int H = cgrad_fx_parts(n);
        chamfer_grad_attack_fx_kernel<<<dim3(B, np * H), CGA_THREADS, cgrad_fx_lds_bytes(n), st>>>(ca, H);
        GA_LAUNCH_CHECK();
        return GEOADV_OK;
    }
    chamfer_grad_attack_kernel<<<dim3(B, np), CGA_THREADS, sizeof(unsigned) * (size_t)ca.P * (ca.P <= CG_TERMS_MAX_P ? 4 : 1), st>>>(ca);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

// backward from the cached forward + Adam
int do_step(geoadv_attack *at, hipStream_t st) {
    const DeviceAE &A = at->ae->d;
    const int B = at->B, n = at->n;
    const bool adv_chamfer = at->cfg.loss_adv_type == GEOADV_LOSS_ADV_CHAMFER;
    const bool dist_chamfer = at->cfg.loss_dist_type == GEOADV_LOSS_DIST_CHAMFER;
    const CGradProblem p_recon{at->recon, at->gt, at->ir1, at->ir2, at->g_recon, nullptr, nullptr, 0.f};
    const CGradProblem p_dist{at->adv, at->x, at->ia1, at->ia2, at->g_dist, at->w, at->jstar, at->cfg.max_point_dist_weight};
    if (!at->cgrad_done) {   // (normally already produced by the forward's loss launch)
        ProfScope ps(at, GEOADV_PROF_LOSS_GRAD, st);
        CGradProblem pr[2];
        int np = 0;
        if (adv_chamfer) pr[np++] = p_recon;
        if (dist_chamfer) pr[np++] = p_dist;
        if (np)
            if (int rc = launch_cgrad(pr, np, B, n, st)) return rc;
    }
    at->cgrad_done = false;
    if (adv_chamfer && at->emd_temp) {   // d(emd_weight * cost / n)/d recon, match held constant: emd_g1 came with the cached forward
        ProfScope ps(at, GEOADV_PROF_LOSS_GRAD, st);
        const size_t total = (size_t)B * n * 3;
        axpy_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(at->g_recon, at->emd_g1, at->cfg.emd_weight / (float)n, total);
        GA_LAUNCH_CHECK();
    }
    if (adv_chamfer) {
        ProfScope ps(at, GEOADV_PROF_DECODER_BWD, st);
        if (at->jac_valid) {
            // the tail applies the Jacobian; the dense recomputing backward for clouds with a tied pool maximum shares its launch
            // (decoder_tail_dense_kernel) and waits, for such clouds only, on the tail block's flag
            // -- as long as the launch's dense blocks (2 n / 32, a CU's LDS each) and tail blocks all fit the chip at once: a
            // dense block spins on a flag only a tail block raises, and nothing orders their dispatch in general.  Larger
            // clouds take the two plain launches (tail with the Jacobian's apply, then the dense blocks: + one boundary).
            if (B + 2 * cdiv(n, 32) > at->cus) {
                if (int rc = launch_decoder_bwd(A, B, at->g_recon, at->fs.d1, at->fs.d2, at->dec_partial, at->dz, st, at->fs.crit, at->jac,
                                                at->fs.dense, at->g_enc)) return rc;
                if (int rc = launch_encoder_bwd_dense(A, B, at->adv, at->fs.z, at->fs.zcnt, at->dz, at->fs.dense, at->g_enc, st)) return rc;
            } else {
            if (int rc = launch_decoder_fc2_bwd(A, B, at->g_recon, at->dec_partial, st)) return rc;
            TailDenseArgs ta;
            ta.batch = B; ta.chunks = decoder_bwd_chunks(A); ta.n = n;
            ta.partial = at->dec_partial; ta.d1 = at->fs.d1; ta.d2 = at->fs.d2; ta.dz = at->dz;
            ta.ja = JacApply{at->fs.crit, at->jac, at->fs.dense, at->g_enc, n};
            ta.adv = at->adv; ta.z = at->fs.z; ta.zcnt = at->fs.zcnt; ta.dense_flag = at->fs.dense; ta.g_enc = at->g_enc;
            ta.ready = at->tail_ready; ta.epoch = ++at->tail_epoch; ta.spin_timeout = reinterpret_cast<int *>(at->tail_ready + B);
            if (int rc = launch_decoder_tail_dense(A, ta, st)) return rc;
            }
        } else if (int rc = launch_decoder_bwd(A, B, at->g_recon, at->fs.d1, at->fs.d2, at->dec_partial, at->dz, st)) return rc;
    }
    if (!(adv_chamfer && at->jac_valid)) {
        ProfScope ps(at, GEOADV_PROF_ENCODER_BWD, st);
        if (int rc = launch_encoder_bwd(A, B, at->adv, at->fs.crit, at->fs.z, at->fs.zcnt, at->dz, at->fs.dense, at->g_enc, at->masks, st))
            return rc;
    }
    {
        const float beta1 = 0.9f, beta2 = 0.999f;
        const float alpha = at->cfg.learning_rate * sqrtf(1.0f - at->beta2_pow) / (1.0f - at->beta1_pow);
        if (at->fuse_adam && dist_chamfer) {
            // every element's update only needs that element: the next forward's point loaders do it on their way in (one
            // launch less per iteration; attack_run always follows a step with a forward)
            at->pending = FusedAdam{at->pert, at->m, at->v, at->g_enc, at->g_dist, at->grad_last, alpha, 1.0f - beta1, 1.0f - beta2, 1e-8f};
            at->adam_pending = true;
        } else {
            ProfScope ps(at, GEOADV_PROF_ADAM, st);
            AdamArgs aa;
            aa.n = n; aa.B = B; aa.pert = at->pert; aa.m = at->m; aa.v = at->v; aa.g_enc = at->g_enc;
            aa.g_dist = dist_chamfer ? at->g_dist : nullptr; aa.w = at->w; aa.losses = at->losses; aa.jstar = at->jstar;
            aa.loss_dist_type = at->cfg.loss_dist_type; aa.mp_pert_w = at->cfg.max_point_pert_weight;
            aa.alpha = alpha;
            aa.one_minus_b1 = 1.0f - beta1; aa.one_minus_b2 = 1.0f - beta2; aa.eps = 1e-8f;
            aa.grad_out = at->grad_last; aa.x = at->x; aa.adv_out = at->adv;
            const size_t total = (size_t)B * n * 3;
            adam_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(aa);
            GA_LAUNCH_CHECK();
        }
        at->beta1_pow *= beta1;
        at->beta2_pow *= beta2;
    }
    at->fwd_valid = false;
    at->adv_valid = !at->adam_pending;
    return GEOADV_OK;
}

}  // namespace

namespace geoadv {   // the same gradient for the training step (train.hip)
int launch_chamfer_grad(const CGradProblem *pr, int np, int B, int n, hipStream_t st) { return launch_cgrad(pr, np, B, n, st); }
}

extern "C" int geoadv_chamfer_per_pc(int b, int n, int m, const float *dist1, const float *dist2, float *out, void *stream) {
    GA_REQUIRE(b >= 0 && n >= 1 && m >= 1, "chamfer_per_pc: bad dimensions (b=%d n=%d m=%d)", b, n, m);
    if (b == 0) return GEOADV_OK;
    GA_REQUIRE(dist1 && dist2 && out, "chamfer_per_pc: null pointer");
    chamfer_per_pc_kernel<<<b, 256, 0, as_stream(stream)>>>(n, m, dist1, dist2, out);
    GA_LAUNCH_CHECK();
    return GEOADV_OK;
}

extern "C" int geoadv_attack_create(geoadv_attack **out, const geoadv_ae *ae, const geoadv_attack_config *cfg) {
    GA_REQUIRE(out && ae && cfg, "attack_create: null argument");
    GA_REQUIRE(cfg->batch >= 1 && cfg->batch <= 65535, "attack_create: batch %d out of range", cfg->batch);
    GA_REQUIRE(cfg->loss_adv_type == GEOADV_LOSS_ADV_CHAMFER || cfg->loss_adv_type == GEOADV_LOSS_ADV_LATENT,
               "attack_create: unknown loss_adv_type %d", cfg->loss_adv_type);
    GA_REQUIRE(cfg->loss_dist_type == GEOADV_LOSS_DIST_CHAMFER || cfg->loss_dist_type == GEOADV_LOSS_DIST_PERT,
               "attack_create: unknown loss_dist_type %d", cfg->loss_dist_type);
    GA_REQUIRE(cfg->emd_weight >= 0.f, "attack_create: emd_weight must be >= 0");
    GA_REQUIRE((cfg->emd_weight_mode & ~GEOADV_EMD_DENSE_LEVELS) == GEOADV_EMD_FAST || (cfg->emd_weight_mode & ~GEOADV_EMD_DENSE_LEVELS) == GEOADV_EMD_REFERENCE,
               "attack_create: unknown emd_weight_mode %d", cfg->emd_weight_mode);
    GA_REQUIRE(cfg->all_pairs_source_dist >= 0 && cfg->all_pairs_source_dist <= 2, "attack_create: all_pairs_source_dist must be 0, 1 or 2");
    GA_REQUIRE(cfg->encoder_backward >= GEOADV_ENC_BWD_AUTO && cfg->encoder_backward <= GEOADV_ENC_BWD_JACOBIAN,
               "attack_create: unknown encoder_backward %d", cfg->encoder_backward);
    GA_REQUIRE(cfg->chamfer_kernel >= GEOADV_CHAMFER_AUTO && cfg->chamfer_kernel <= GEOADV_CHAMFER_SYMMETRIC,
               "attack_create: unknown chamfer_kernel %d", cfg->chamfer_kernel);
    GA_REQUIRE(cfg->emd_weight == 0.f || cfg->loss_adv_type == GEOADV_LOSS_ADV_CHAMFER,
               "attack_create: emd_weight needs the output-space attack (loss_adv_type chamfer)");
    geoadv_attack *at = new geoadv_attack();
    at->ae = ae; at->cfg = *cfg; at->B = cfg->batch; at->n = ae->d.n_points;
    const size_t B = at->B, n = at->n, bn3 = B * n * 3, bn = B * n;
    const int chunks = decoder_bwd_chunks(ae->d);
    const size_t fs_bytes = carve_forward_scratch(nullptr, at->B, at->n).bytes;
    size_t total = 0;
    auto need = [&](size_t bytes) { total += rup(bytes, 256); };
    for (int i = 0; i < 2; ++i) need(4 * bn3);            // x, gt
    need(4 * B * 128); need(4 * B);                       // tz, w
    for (int i = 0; i < 3; ++i) need(4 * bn3);            // pert, m, v
    for (int i = 0; i < 6; ++i) need(4 * bn3);            // adv, recon, g_recon, g_dist, g_enc, grad_last
    for (int i = 0; i < 8; ++i) need(4 * bn);             // r1 r2 a1 a2 + 4 idx
    need(fs_bytes);
    need(4 * B * 128); need(4 * (size_t)chunks * B * 256);
    need(4 * 8 * B); need(4 * 2 * B);
    need(4 * B); need(4 * B * 4); need(4 * bn3); need(4 * bn3);
    const size_t sym_floats = chamfer_sym_workspace_floats(2, at->B, at->n, at->n);
    need(4 * sym_floats);
    need(8 * 2 * bn);                                     // row64
    const size_t mask_words = cfg->recompute_backward ? 0 : (size_t)encoder_mask_words() * bn;   // ReLU masks of the cached forward
    need(4 * mask_words);
    const bool use_jac = mask_words != 0 && cfg->loss_adv_type == GEOADV_LOSS_ADV_CHAMFER && cfg->encoder_backward != GEOADV_ENC_BWD_MASKED;
    need(use_jac ? 4 * B * 128 * 3 : 0);
    need(4 * (B + 1));
    need(4 * 8 * B); need(4 * 8 * B);                     // need_adv[2]
    need(4 * 6 * B);                                      // x_box
    const bool emd = cfg->emd_weight > 0.f;
    const size_t emd_temp_f = emd ? geoadv_emd_cost_grad1_temp_floats(at->B, at->n, at->n) : 0;
    if (emd) { need(4 * emd_temp_f + 8); need(4 * B); need(4 * bn3); }
    at->arena_bytes = total;
    if (hipMalloc(&at->arena, total) != hipSuccess) {
        delete at;
        set_error("attack_create: hipMalloc of %zu bytes failed", total);
        return GEOADV_ENOMEM;
    }
    // the clear runs on the NULL stream and returns before it is done; callers' streams may be non-blocking ones (torch's are),
    // which the null stream does not order against -- without the wait a busy GPU (another host thread's work) lets the clear
    // land AFTER the caller's first uploads into the arena (seen as 1-3 % of two-thread runs computing on zeroed inputs)
    if (hipMemsetAsync(at->arena, 0, total, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipFree(at->arena);
        delete at;
        set_error("attack_create: clearing %zu bytes failed", total);
        return GEOADV_EHIP;
    }
    char *p = static_cast<char *>(at->arena);
    auto take = [&](size_t bytes) { char *q = p; p += rup(bytes, 256); return q; };
    auto F = [&](size_t bytes) { return reinterpret_cast<float *>(take(bytes)); };
    auto I = [&](size_t bytes) { return reinterpret_cast<int *>(take(bytes)); };
    at->x = F(4 * bn3); at->gt = F(4 * bn3); at->tz = F(4 * B * 128); at->w = F(4 * B);
    at->pert = F(4 * bn3); at->m = F(4 * bn3); at->v = F(4 * bn3);
    at->adv = F(4 * bn3); at->recon = F(4 * bn3); at->g_recon = F(4 * bn3); at->g_dist = F(4 * bn3);
    at->g_enc = F(4 * bn3); at->grad_last = F(4 * bn3);
    at->r1 = F(4 * bn); at->r2 = F(4 * bn); at->a1 = F(4 * bn); at->a2 = F(4 * bn);
    at->ir1 = I(4 * bn); at->ir2 = I(4 * bn); at->ia1 = I(4 * bn); at->ia2 = I(4 * bn);
    at->fs = carve_forward_scratch(take(fs_bytes), at->B, at->n);
    at->dz = F(4 * B * 128); at->dec_partial = F(4 * (size_t)chunks * B * 256);
    at->losses = F(4 * 8 * B); at->jstar = I(4 * 2 * B);
    at->best_err = F(4 * B); at->best_metrics = F(4 * B * 4); at->best_adv = F(4 * bn3); at->best_recon = F(4 * bn3);
    at->sym_ws = F(4 * sym_floats);
    at->row64 = reinterpret_cast<unsigned long long *>(take(8 * 2 * bn));
    at->masks = mask_words ? reinterpret_cast<unsigned *>(take(4 * mask_words)) : nullptr;
    at->jac = use_jac ? F(4 * B * 128 * 3) : nullptr;
    at->jac_valid = false;
    at->tail_ready = reinterpret_cast<unsigned *>(take(4 * (B + 1)));      // (the arena is zeroed above)
    at->tail_epoch = 0;
    at->loss_done = reinterpret_cast<unsigned *>(take(4 * (size_t)B));
    at->loss_target = 0;
    {
        int dev = 0, cus = 0;                               // unknown => 0: always the two plain launches
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) at->cus = cus;
        else at->cus = 0;
    }
    at->need_adv[0] = I(4 * 8 * B); at->need_adv[1] = I(4 * 8 * B);
    at->grid_calls = 0;
    at->x_box = F(4 * 6 * B);
    // Tiny batches (< 10 K points: B <= 4 at N = 2048): the search's own latency (~11-14 us per workgroup beside a 6 us
    // latent_decode) costs what the two extra all-pairs scans cost on a mostly idle chip -- measured 0.0673 / 0.0725 ms at
    // B = 1 and 0.0758 / 0.0767 at B = 4 without / with it, 0.0977 / 0.0929 at B = 8 -- so it is only used above that
    // (at B = 5 the two-scan launch needs a third round of workgroups: 0.0908 without, 0.0846 with the search + symmetric scan).
    // (all_pairs_source_dist 2 = the search whatever the size: the parity tests' small shapes)
    at->chamfer_prune = cfg->all_pairs_source_dist == 2 || (cfg->all_pairs_source_dist == 0 && (long)at->B * at->n >= GEOADV_SMALL_BATCH_POINTS);
    {
        // The symmetric scan against the public op's plain scans in ONE launch (both directions of (recon, target), and of (adv,
        // source) only for clouds the grid search handed back).  Until round 4 it needed a finish launch and won from B = 5 on
        // (ms per iteration, plain / symmetric: B = 4: 0.0749 / 0.0743, 5: 0.0909 / 0.0846, 8: 0.0935 / 0.0925, 32: 0.1913 / 0.1773).
        // Round 5: the column minima are resolved inside the scan and the row minima reach the loss launch as per-slice partials or
        // -- narrow slices, i.e. these small batches -- as packed words folded by 64-bit atomic minima, so the symmetric form is ONE
        // Chamfer launch at every size and carries the encoder's pool Jacobian (no masked backward launch): plain / symmetric
        // B = 2: 0.0675 / 0.0647, 3: 0.0736 / 0.0657, 4: 0.0743 / 0.0660, 5: 0.0917 / 0.0776, 8: 0.0933 / 0.0792, 16: 0.1262 / 0.1070
        // (profiles/r05_small_batch_paths.jsonl).  Same bits either way.
        at->chamfer_sym = cfg->chamfer_kernel == GEOADV_CHAMFER_AUTO ? (long)at->B * at->n >= GEOADV_SYM_MIN_POINTS
                                                                     : cfg->chamfer_kernel == GEOADV_CHAMFER_SYMMETRIC;
    }
    at->emd_temp = at->emd_cost = at->emd_g1 = nullptr;
    if (emd) { at->emd_temp = F(4 * emd_temp_f + 8); at->emd_cost = F(4 * B); at->emd_g1 = F(4 * bn3); }
    at->cgrad_done = false;
    at->fuse_adam = cfg->separate_adam == 0;
    at->adam_pending = false;
    at->beta1_pow = 0.9f; at->beta2_pow = 0.999f;      // TF: beta*_power variables start at beta*
    at->fwd_valid = false; at->adv_valid = false;
    at->prof_mask = 0; at->ev_used = 0; at->prof_stream = nullptr; at->prof_stride = 1; at->markers = false;
    for (int i = 0; i < GEOADV_PROF_COUNT; ++i) at->prof_seen[i] = 0;
    for (int i = 0; i < GEOADV_PROF_COUNT; ++i) { at->prof_ms[i] = 0; at->prof_n[i] = 0; }
    static DeviceOnce attr;
    if (int rc = attr.run([]() -> int {
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_grad_attack_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_grad_attack_fx_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 24 * CG_FX_MAX_N));
            GA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(loss_cgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 24 * CG_FX_MAX_N));
            return GEOADV_OK;
        })) { geoadv_attack_destroy(at); return rc; }
    at->host_loss = false;
    at->host_loss_force = cfg->loss_in_scan == 2;
    if (cfg->loss_in_scan != 1 && at->chamfer_sym) {           // does this device deal workgroup i to XCD i % 8?
        unsigned *d = reinterpret_cast<unsigned *>(at->sym_ws), h[64];
        xcd_probe_kernel<<<64, 64>>>(d);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            bool ok = true;
            for (int i = 8; i < 64; ++i) ok = ok && h[i] == h[i & 7];
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < i; ++j) ok = ok && h[i] != h[j];
            at->host_loss = ok;
        }
        (void)hipGetLastError();
    }
    ae->attack_refs.fetch_add(1);
    at->counted = true;
    *out = at;
    return GEOADV_OK;
}

extern "C" void geoadv_attack_destroy(geoadv_attack *at) {
    if (!at) return;
    if (at->counted) at->ae->attack_refs.fetch_sub(1);
    for (auto e : at->ev) (void)hipEventDestroy(e);
    (void)hipFree(at->arena);
    delete at;
}

extern "C" int geoadv_attack_set_inputs(geoadv_attack *at, const float *source_pc, const float *target_pc,
                                        const float *target_latent, const float *dist_weight, void *stream) {
    GA_REQUIRE(at && source_pc && target_pc && dist_weight, "attack_set_inputs: null argument");
    GA_REQUIRE(target_latent || at->cfg.loss_adv_type != GEOADV_LOSS_ADV_LATENT,
               "attack_set_inputs: target_latent is required for loss_adv_type 'latent'");
    hipStream_t st = as_stream(stream);
    const size_t bn3 = (size_t)at->B * at->n * 3;
    if (at->adam_pending)    // a step that has not reached its forward yet: apply it to pert (with the old clouds) before they change
        if (int rc = do_forward(at, nullptr, 0, st)) return rc;
    GA_HIP(hipMemcpyAsync(at->x, source_pc, 4 * bn3, hipMemcpyDeviceToDevice, st));
    GA_HIP(hipMemcpyAsync(at->gt, target_pc, 4 * bn3, hipMemcpyDeviceToDevice, st));
    if (target_latent) GA_HIP(hipMemcpyAsync(at->tz, target_latent, 4 * (size_t)at->B * 128, hipMemcpyDeviceToDevice, st));
    GA_HIP(hipMemcpyAsync(at->w, dist_weight, 4 * (size_t)at->B, hipMemcpyDeviceToDevice, st));
    if (int rc = launch_chamfer_grid_box(at->x, at->B, at->n, at->x_box, st)) return rc;   // grid of the paired nn search
    for (int i = 0; i < 2; ++i)                                                            // new clouds: every verdict is open again
        GA_HIP(hipMemsetAsync(at->need_adv[i], 0, sizeof(int) * 8 * (size_t)at->B, st));
    GA_HIP(hipMemsetAsync(at->tail_ready + at->B, 0, sizeof(unsigned), st));               // a new run starts without a failure on record
    at->fwd_valid = false; at->adv_valid = false;
    return GEOADV_OK;
}

extern "C" int geoadv_attack_init_pert(geoadv_attack *at, const float *init_pert, int reset_optimizer, void *stream) {
    GA_REQUIRE(at && init_pert, "attack_init_pert: null argument");
    hipStream_t st = as_stream(stream);
    const size_t bn3 = (size_t)at->B * at->n * 3;
    at->adam_pending = false;                          // (a step that never reached its forward is dropped with the old pert)
    GA_HIP(hipMemsetAsync(at->tail_ready + at->B, 0, sizeof(unsigned), st));
    GA_HIP(hipMemcpyAsync(at->pert, init_pert, 4 * bn3, hipMemcpyDeviceToDevice, st));
    if (reset_optimizer) {
        GA_HIP(hipMemsetAsync(at->m, 0, 4 * bn3, st));
        GA_HIP(hipMemsetAsync(at->v, 0, 4 * bn3, st));
        at->beta1_pow = 0.9f; at->beta2_pow = 0.999f;
    }
    // adv_ae.py:197-200: best error 1e10, metrics / clouds zero
    fill_kernel<<<(at->B + 255) / 256, 256, 0, st>>>(at->best_err, 1e10f, (size_t)at->B);
    GA_LAUNCH_CHECK();
    GA_HIP(hipMemsetAsync(at->best_metrics, 0, 4 * (size_t)at->B * 4, st));
    GA_HIP(hipMemsetAsync(at->best_adv, 0, 4 * bn3, st));
    GA_HIP(hipMemsetAsync(at->best_recon, 0, 4 * bn3, st));
    at->fwd_valid = false; at->adv_valid = false;
    return GEOADV_OK;
}

extern "C" int geoadv_attack_run(geoadv_attack *at, int first_iteration, int iterations, int thresh,
                                 float *metrics_hist, void *stream) {
    GA_REQUIRE(at && first_iteration >= 0 && iterations >= 0, "attack_run: bad arguments");
    hipStream_t st = as_stream(stream);
    at->prof_stream = st;
    if (iterations == 0) return GEOADV_OK;
    if (!at->fwd_valid)
        if (int rc = do_forward(at, nullptr, 0, st)) return rc;
    for (int i = 0; i < iterations; ++i) {
        if (int rc = do_step(at, st)) return rc;
        const int it = first_iteration + i;                       // 0-based index within the dist-weight run
        const int keep = (it + 1) >= thresh ? 1 : 0;              // adv_ae.py:234
        float *slot = metrics_hist ? metrics_hist + (size_t)i * 6 * at->B : nullptr;
        if (int rc = do_forward(at, slot, keep, st)) return rc;
    }
    return GEOADV_OK;
}

extern "C" int geoadv_attack_get_best(geoadv_attack *at, const float *target_ae_loss_ref, float *metrics, float *adv,
                                      float *recon, void *stream) {
    GA_REQUIRE(at, "attack_get_best: null handle");
    hipStream_t st = as_stream(stream);
    const size_t bn3 = (size_t)at->B * at->n * 3;
    if (metrics) {
        GA_REQUIRE(target_ae_loss_ref, "attack_get_best: target_ae_loss_ref is required for the metrics");
        best_metrics_kernel<<<(at->B + 63) / 64, 64, 0, st>>>(at->B, at->best_metrics, at->best_err, target_ae_loss_ref, metrics,
                                                              reinterpret_cast<const int *>(at->tail_ready + at->B));
        GA_LAUNCH_CHECK();
    }
    if (adv) GA_HIP(hipMemcpyAsync(adv, at->best_adv, 4 * bn3, hipMemcpyDeviceToDevice, st));
    if (recon) GA_HIP(hipMemcpyAsync(recon, at->best_recon, 4 * bn3, hipMemcpyDeviceToDevice, st));
    return GEOADV_OK;
}

extern "C" int geoadv_attack_set_source_search(geoadv_attack *at, int on) {
    GA_REQUIRE(at, "attack_set_source_search: null handle");
    at->chamfer_prune = on != 0;           // read by every forward (do_forward); the verdict flags keep their state
    return GEOADV_OK;
}

extern "C" int geoadv_attack_status(geoadv_attack *at, void *stream) {
    GA_REQUIRE(at, "attack_status: null handle");
    hipStream_t st = as_stream(stream);
    int failed = 0;
    GA_HIP(hipMemcpyAsync(&failed, at->tail_ready + at->B, sizeof(int), hipMemcpyDeviceToHost, st));
    GA_HIP(hipStreamSynchronize(st));
    if (failed) {
        set_error("attack: an in-launch hand-off (decoder backward tail -> dense encoder backward of a cloud with a tied pool "
                  "maximum) timed out since the last set_inputs / init_pert; iterations after it used a stale latent gradient -- "
                  "the results of this run are invalid");
        return GEOADV_EHIP;
    }
    return ae_range_check(at->ae, st, "attack_status");        // the victim's f16x2 range guard (encoder_x3.h)
}

extern "C" int geoadv_attack_peek(geoadv_attack *at, float *pert, float *adv, float *recon, float *latent, float *grad,
                                  int *idx_r1, int *idx_r2, int *idx_a1, int *idx_a2, void *stream) {
    GA_REQUIRE(at, "attack_peek: null handle");
    hipStream_t st = as_stream(stream);
    const size_t bn3 = 4 * (size_t)at->B * at->n * 3, bn = 4 * (size_t)at->B * at->n;
    // a step whose Adam update is still waiting for the next forward's point loaders (fused Adam) has not touched pert /
    // grad_last yet: run that forward first, whatever is asked for
    if (at->adam_pending || (!at->fwd_valid && (adv || recon || latent || idx_r1 || idx_r2 || idx_a1 || idx_a2)))
        if (int rc = do_forward(at, nullptr, 0, st)) return rc;
    if (pert) GA_HIP(hipMemcpyAsync(pert, at->pert, bn3, hipMemcpyDeviceToDevice, st));
    if (adv) GA_HIP(hipMemcpyAsync(adv, at->adv, bn3, hipMemcpyDeviceToDevice, st));
    if (recon) GA_HIP(hipMemcpyAsync(recon, at->recon, bn3, hipMemcpyDeviceToDevice, st));
    if (latent) GA_HIP(hipMemcpyAsync(latent, at->fs.z, 4 * (size_t)at->B * 128, hipMemcpyDeviceToDevice, st));
    if (grad) GA_HIP(hipMemcpyAsync(grad, at->grad_last, bn3, hipMemcpyDeviceToDevice, st));
    if (idx_r1) GA_HIP(hipMemcpyAsync(idx_r1, at->ir1, bn, hipMemcpyDeviceToDevice, st));
    if (idx_r2) GA_HIP(hipMemcpyAsync(idx_r2, at->ir2, bn, hipMemcpyDeviceToDevice, st));
    if (idx_a1) GA_HIP(hipMemcpyAsync(idx_a1, at->ia1, bn, hipMemcpyDeviceToDevice, st));
    if (idx_a2) GA_HIP(hipMemcpyAsync(idx_a2, at->ia2, bn, hipMemcpyDeviceToDevice, st));
    return GEOADV_OK;
}

extern "C" int geoadv_attack_search_state(geoadv_attack *at, int *searched, int *handed_back, void *stream) {
    GA_REQUIRE(at && searched && handed_back, "attack_search_state: null argument");
    const bool pruned = at->chamfer_prune && chamfer_grid_supports(at->n, at->n);
    *searched = pruned ? 1 : 0;
    *handed_back = 0;
    if (!pruned) return GEOADV_OK;
    hipStream_t st = as_stream(stream);
    const bool two = chamfer_grid_rides(at->n) && at->chamfer_sym;                 // (two flag arrays: the last call wrote this one)
    const int *flags = at->need_adv[two ? at->grid_calls & 1 : 0];
    std::vector<int> h(8 * (size_t)at->B);
    GA_HIP(hipMemcpyAsync(h.data(), flags, sizeof(int) * h.size(), hipMemcpyDeviceToHost, st));
    GA_HIP(hipStreamSynchronize(st));
    for (int c = 0; c < at->B; ++c) {
        int any = 0;
        for (int k = 0; k < 8; ++k) any |= h[8 * (size_t)c + k];
        *handed_back += any ? 1 : 0;
    }
    return GEOADV_OK;
}

extern "C" int geoadv_attack_profile(geoadv_attack *at, int enable) {
    GA_REQUIRE(at, "attack_profile: null handle");
    if (enable && at->ev.empty()) {
        at->ev.resize(4096);
        for (auto &e : at->ev) GA_HIP(hipEventCreate(&e));
    }
    if (at->prof_mask) { if (int rc = prof_flush(at)) return rc; }
    at->prof_mask = (unsigned)enable;
    if (enable) for (int i = 0; i < GEOADV_PROF_COUNT; ++i) { at->prof_ms[i] = 0; at->prof_n[i] = 0; }
    return GEOADV_OK;
}

extern "C" int geoadv_attack_markers(geoadv_attack *at, int enable) {
    GA_REQUIRE(at, "attack_markers: null handle");
    GA_REQUIRE(!enable || roctx().push, "attack_markers: libroctx64.so not found");
    at->markers = enable != 0;
    return GEOADV_OK;
}

extern "C" int geoadv_attack_profile_stride(geoadv_attack *at, int stride) {
    GA_REQUIRE(at && stride >= 1, "attack_profile_stride: bad arguments");
    at->prof_stride = stride;
    return GEOADV_OK;
}

extern "C" int geoadv_attack_profile_read(geoadv_attack *at, int which, int *launches, float *total_ms) {
    GA_REQUIRE(at && which >= 0 && which < GEOADV_PROF_COUNT && launches && total_ms, "attack_profile_read: bad arguments");
    if (int rc = prof_flush(at)) return rc;
    *launches = at->prof_n[which];
    *total_ms = (float)at->prof_ms[which];
    return GEOADV_OK;
}
GA_STAMPS_GETTER(geoadv_debug_stamps_attack)
