/*
 * geoadv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's custom-op arithmetic for the
 * geometric-adversarial attack path (itailang/geometric_adv).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (geometric_adv_amd) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against golden vectors produced in the build container by the reference's own
 * CPU functions (oracle/build_ref.sh -> oracle/_ref/libgeoadv_ref.so ->
 * oracle/make_golden.py -> tests/golden/*.npz).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  The
 * -ffp-contract=off matters: the reference is built by g++ -O2 for baseline
 * x86-64 (no FMA instructions), so every float product and sum below is rounded
 * on its own.
 *
 * Citations are file:line in /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------
 * Chamfer nearest neighbour, one direction.
 * Follows external/structural_losses/tf_nndistance.cpp:21-43 (nnsearch):
 *   - offsets are target-minus-query, squared and summed left to right in float
 *     (three roundings for the products, two for the sums),
 *   - the first candidate is always taken, later ones only on strict '<'
 *     => the lowest index wins ties,
 *   - the running best is kept in a double but only ever holds float values.
 * ------------------------------------------------------------------------- */
void oracle_nn_search(int b, int n, int m, const float *query, const float *target,
                      float *dist, int *idx)
{
#ifdef GEOADV_ORACLE_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int c = 0; c < b; ++c) {
        const float *q = query + (size_t)c * n * 3;
        const float *t = target + (size_t)c * m * 3;
        for (int j = 0; j < n; ++j) {
            const float qx = q[3 * j], qy = q[3 * j + 1], qz = q[3 * j + 2];
            double best = 0.0;
            int arg = 0;
            for (int k = 0; k < m; ++k) {
                const float dx = t[3 * k] - qx;
                const float dy = t[3 * k + 1] - qy;
                const float dz = t[3 * k + 2] - qz;
                const float sq = dx * dx + dy * dy + dz * dz; /* float arithmetic, widened after */
                const double d = sq;
                if (k == 0 || d < best) { best = d; arg = k; }
            }
            dist[(size_t)c * n + j] = (float)best;
            idx[(size_t)c * n + j] = arg;
        }
    }
}

/* NnDistanceOp::Compute = two searches (tf_nndistance.cpp:79-80). */
void oracle_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2,
                        float *dist1, int *idx1, float *dist2, int *idx2)
{
    oracle_nn_search(b, n, m, xyz1, xyz2, dist1, idx1);
    oracle_nn_search(b, m, n, xyz2, xyz1, dist2, idx2);
}

/* ---------------------------------------------------------------------------
 * Chamfer gradient.  Follows tf_nndistance.cpp:126-163: zero both outputs, then
 * per cloud first the xyz1 -> xyz2 matches in ascending j, then the xyz2 -> xyz1
 * matches in ascending j; g = 2*grad_dist; the own point gets +g*(p-q), the matched
 * point gets -(g*(p-q)).  Accumulation order is therefore fully defined.
 * ------------------------------------------------------------------------- */
static void scatter_half(int cnt, int other_cnt, const float *p, const float *q, const float *gd,
                         const int *match, float *gp, float *gq)
{
    (void)other_cnt;
    for (int j = 0; j < cnt; ++j) {
        const int k = match[j];
        const float g = gd[j] * 2;
        for (int a = 0; a < 3; ++a) {
            const float t = g * (p[3 * j + a] - q[3 * k + a]);
            gp[3 * j + a] += t;
            gq[3 * k + a] -= t;
        }
    }
}

void oracle_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                             const float *grad_dist1, const int *idx1,
                             const float *grad_dist2, const int *idx2,
                             float *grad_xyz1, float *grad_xyz2)
{
    memset(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3);
    for (int c = 0; c < b; ++c) {
        const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + (size_t)c * m * 3;
        float *gp = grad_xyz1 + (size_t)c * n * 3, *gq = grad_xyz2 + (size_t)c * m * 3;
        scatter_half(n, m, p, q, grad_dist1 + (size_t)c * n, idx1 + (size_t)c * n, gp, gq);
        scatter_half(m, n, q, p, grad_dist2 + (size_t)c * m, idx2 + (size_t)c * m, gq, gp);
    }
}

/* ---------------------------------------------------------------------------
 * approx-EMD soft matching, CPU variant.
 * Follows external/structural_losses/tf_approxmatch.cpp:23-84:
 *   - 11 temperature levels j = 8 .. -2, level = -4^j (powf in float), 0 for j == -2,
 *   - per-point capacities start at max(n,m)/n and max(n,m)/m (integer division),
 *   - all bookkeeping in double; the kernel value is expf() of the double exponent
 *     narrowed to float,
 *   - match is laid out (n, m): match[k*m + l] for xyz1 point k and xyz2 point l
 *     (the op declares the output shape (b,m,n), tf_approxmatch.cpp:194 -- the CPU
 *     function nevertheless writes it row-major over k; we keep the CPU layout).
 * ------------------------------------------------------------------------- */
void oracle_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match)
{
    const int big = n > m ? n : m;
    double *capl = (double *)malloc(sizeof(double) * n);
    double *capr = (double *)malloc(sizeof(double) * m);
    double *w = (double *)malloc(sizeof(double) * (size_t)n * m);
    double *colsum = (double *)malloc(sizeof(double) * m);
    double *colused = (double *)malloc(sizeof(double) * m);
    for (int c = 0; c < b; ++c) {
        const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + (size_t)c * m * 3;
        float *out = match + (size_t)c * n * m;
        for (int k = 0; k < n; ++k) capl[k] = (double)(big / n);
        for (int l = 0; l < m; ++l) capr[l] = (double)(big / m);
        for (size_t e = 0; e < (size_t)n * m; ++e) out[e] = 0;
        for (int j = 8; j >= -2; --j) {
            double level = -powf(4.0f, (float)j);
            if (j == -2) level = 0;
            for (int k = 0; k < n; ++k) {
                const double x1 = p[3 * k], y1 = p[3 * k + 1], z1 = p[3 * k + 2];
                for (int l = 0; l < m; ++l) {
                    const double x2 = q[3 * l], y2 = q[3 * l + 1], z2 = q[3 * l + 2];
                    const double e = level * ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                    w[(size_t)k * m + l] = expf((float)e) * capr[l];
                }
            }
            for (int l = 0; l < m; ++l) colsum[l] = 1e-9;
            for (int k = 0; k < n; ++k) {
                double rs = 1e-9;
                for (int l = 0; l < m; ++l) rs += w[(size_t)k * m + l];
                for (int l = 0; l < m; ++l) w[(size_t)k * m + l] = w[(size_t)k * m + l] / rs * capl[k];
                for (int l = 0; l < m; ++l) colsum[l] += w[(size_t)k * m + l];
            }
            for (int l = 0; l < m; ++l) {
                const double r = capr[l] / colsum[l];
                colsum[l] = r < 1.0 ? r : 1.0;
            }
            for (int l = 0; l < m; ++l) colused[l] = 0;
            for (int k = 0; k < n; ++k) {
                double rs = 0;
                for (int l = 0; l < m; ++l) {
                    w[(size_t)k * m + l] *= colsum[l];
                    rs += w[(size_t)k * m + l];
                    colused[l] += w[(size_t)k * m + l];
                }
                const double left = capl[k] - rs;
                capl[k] = left > 0.0 ? left : 0.0;
            }
            for (size_t e = 0; e < (size_t)n * m; ++e) out[e] += w[e];
            for (int l = 0; l < m; ++l) {
                const double left = capr[l] - colused[l];
                capr[l] = left > 0.0 ? left : 0.0;
            }
        }
    }
    free(capl); free(capr); free(w); free(colsum); free(colused);
}

/* MatchCost, tf_approxmatch.cpp:85-105: float sqrtf * match, accumulated in double. */
void oracle_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2,
                       const float *match, float *cost)
{
    for (int c = 0; c < b; ++c) {
        const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + (size_t)c * m * 3;
        const float *mt = match + (size_t)c * n * m;
        double acc = 0;
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < m; ++k) {
                const float dx = q[3 * k] - p[3 * j], dy = q[3 * k + 1] - p[3 * j + 1], dz = q[3 * k + 2] - p[3 * j + 2];
                const float d = sqrtf(dx * dx + dy * dy + dz * dz) * mt[(size_t)j * m + k];
                acc += d;
            }
        cost[c] = (float)acc;
    }
}

/* MatchCostGrad, tf_approxmatch.cpp:106-140.  The reference zeroes only the x component of
 * grad1 (:108-109) and accumulates y/z into whatever the output buffer held; TF's allocator
 * does not promise zeros.  The evident intent -- and what the golden vectors pin, by handing
 * the reference a zero-initialised buffer -- is a fully zeroed grad1, which is what we do.
 * Loop order kept: outer over xyz2 points j, inner over xyz1 points k, all in float. */
void oracle_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                            const float *match, float *grad1, float *grad2)
{
    for (int c = 0; c < b; ++c) {
        const float *p = xyz1 + (size_t)c * n * 3, *q = xyz2 + (size_t)c * m * 3;
        const float *mt = match + (size_t)c * n * m;
        float *g1 = grad1 + (size_t)c * n * 3, *g2 = grad2 + (size_t)c * m * 3;
        for (int k = 0; k < 3 * n; ++k) g1[k] = 0;
        for (int j = 0; j < m; ++j) {
            float sx = 0, sy = 0, sz = 0;
            for (int k = 0; k < n; ++k) {
                const float ox = q[3 * j] - p[3 * k], oy = q[3 * j + 1] - p[3 * k + 1], oz = q[3 * j + 2] - p[3 * k + 2];
                float d = sqrtf(ox * ox + oy * oy + oz * oz);
                if (d < 1e-20f) d = 1e-20f;                    /* std::max(d, 1e-20f) */
                const float wgt = mt[(size_t)k * m + j];
                const float gx = wgt * (ox / d), gy = wgt * (oy / d), gz = wgt * (oz / d);
                g1[3 * k] -= gx; g1[3 * k + 1] -= gy; g1[3 * k + 2] -= gz;
                sx += gx; sy += gy; sz += gz;
            }
            g2[3 * j] = sx; g2[3 * j + 1] = sy; g2[3 * j + 2] = sz;
        }
    }
}

/* ---------------------------------------------------------------------------
 * Partial selection sort of each row of a (b, m, n) distance matrix.
 * Follows external/grouping/tf_grouping_g.cu:83-123 (selection_sort_gpu) and its CPU twin
 * external/grouping/test/selection_sort.cpp:20-63: copy row, iota indices, then for
 * s < k find the FIRST minimum of positions [s, n) (strict '<') and swap it into s.
 * The swap moves the displaced element to the minimum's old slot, which is what gives
 * the op its particular order among equal distances.  All n entries of idx/val are
 * written (only the first k are meaningful).
 * ------------------------------------------------------------------------- */
void oracle_selection_sort(int b, int n, int m, int k, const float *dist, int *idx, float *val)
{
    for (size_t r = 0; r < (size_t)b * m; ++r) {
        const float *src = dist + r * n;
        float *v = val + r * n;
        int *id = idx + r * n;
        for (int s = 0; s < n; ++s) { v[s] = src[s]; id[s] = s; }
        for (int s = 0; s < k && s < n; ++s) {
            int lo = s;
            for (int t = s + 1; t < n; ++t)
                if (v[t] < v[lo]) lo = t;
            if (lo != s) {
                const float tv = v[lo]; v[lo] = v[s]; v[s] = tv;
                const int ti = id[lo]; id[lo] = id[s]; id[s] = ti;
            }
        }
    }
}

/* QueryBallPoint, tf_grouping_g.cu:3-36 / test/query_ball_point.cpp:19-47: the first nsample
 * dataset points with max(sqrt(d2),1e-20) < radius, padded with the first hit; plus the count. */
void oracle_query_ball_point(int b, int n, int m, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx, int *pts_cnt)
{
    for (int c = 0; c < b; ++c) {
        const float *data = xyz1 + (size_t)c * n * 3, *qry = xyz2 + (size_t)c * m * 3;
        for (int j = 0; j < m; ++j) {
            int *row = idx + ((size_t)c * m + j) * nsample;
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                const float dx = qry[3 * j] - data[3 * k], dy = qry[3 * j + 1] - data[3 * k + 1], dz = qry[3 * j + 2] - data[3 * k + 2];
                float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (d < 1e-20f) d = 1e-20f;
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt++] = k;
                }
            }
            if (pts_cnt) pts_cnt[(size_t)c * m + j] = cnt;
        }
    }
}

/* GroupPoint gather / GroupPointGrad scatter-add, tf_grouping_g.cu:40-78. */
void oracle_group_point(int b, int n, int c, int m, int nsample, const float *points,
                        const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (size_t e = 0; e < (size_t)m * nsample; ++e) {
            const int src = idx[(size_t)i * m * nsample + e];
            memcpy(out + ((size_t)i * m * nsample + e) * c, points + ((size_t)i * n + src) * c, sizeof(float) * c);
        }
}

void oracle_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                             const int *idx, float *grad_points)
{
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);   /* tf_grouping.cpp:204 */
    for (int i = 0; i < b; ++i)
        for (size_t e = 0; e < (size_t)m * nsample; ++e) {
            const int dst = idx[(size_t)i * m * nsample + e];
            for (int l = 0; l < c; ++l)
                grad_points[((size_t)i * n + dst) * c + l] += grad_out[((size_t)i * m * nsample + e) * c + l];
        }
}

/* ---------------------------------------------------------------------------
 * knn_point (external/grouping/tf_grouping.py:48-75): dense squared-distance matrix
 * dist[b,q,p] = sum_c (xyz1[b,p,c]-xyz2[b,q,c])^2 (summed left to right), then the
 * selection sort above; val/idx are the first k columns.
 * ------------------------------------------------------------------------- */
void oracle_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2,
                      float *val, int *idx)
{
    float *row = (float *)malloc(sizeof(float) * n);
    float *sv = (float *)malloc(sizeof(float) * n);
    int *si = (int *)malloc(sizeof(int) * n);
    for (int c = 0; c < b; ++c)
        for (int q = 0; q < m; ++q) {
            const float *qp = xyz2 + ((size_t)c * m + q) * 3;
            for (int p = 0; p < n; ++p) {
                const float *dp = xyz1 + ((size_t)c * n + p) * 3;
                const float dx = dp[0] - qp[0], dy = dp[1] - qp[1], dz = dp[2] - qp[2];
                row[p] = dx * dx + dy * dy + dz * dz;
            }
            oracle_selection_sort(1, n, 1, k, row, si, sv);
            for (int s = 0; s < k; ++s) {
                val[((size_t)c * m + q) * k + s] = sv[s];
                idx[((size_t)c * m + q) * k + s] = si[s];
            }
        }
    free(row); free(sv); free(si);
}

/* defender/get_knn_dists_per_point.py:78-81: knn_point(k+1, pc, pc), drop column 0, gather the
 * neighbours, distance = sqrt(sum (neighbour - centre)^2). */
void oracle_knn_dists(int b, int n, int k, const float *pc, float *out)
{
    const int kk = k + 1;
    float *val = (float *)malloc(sizeof(float) * (size_t)n * kk);
    int *idx = (int *)malloc(sizeof(int) * (size_t)n * kk);
    for (int c = 0; c < b; ++c) {
        const float *p = pc + (size_t)c * n * 3;
        oracle_knn_point(1, n, n, kk, p, p, val, idx);
        for (int q = 0; q < n; ++q)
            for (int s = 0; s < k; ++s) {
                const int nb = idx[(size_t)q * kk + s + 1];
                const float dx = p[3 * nb] - p[3 * q], dy = p[3 * nb + 1] - p[3 * q + 1], dz = p[3 * nb + 2] - p[3 * q + 2];
                out[((size_t)c * n + q) * k + s] = sqrtf(dx * dx + dy * dy + dz * dz);
            }
    }
    free(val); free(idx);
}
