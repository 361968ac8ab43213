// Fluid-model LP of the reference (environments/class_FJSSP.py:246-280), solved
// on the host with a dense primal simplex (Dantzig pricing, lexicographic ratio
// test => finite on this highly degenerate LP, deterministic vertex).
//
//   variables   X[m,(r,j)] in [0,1] for every eligible pair (:253-255), t (the
//               model.min auxiliary of :264)
//   maximise    t
//   subject to  t <= (sum_m X[m,k] * (1/p[m][k])) / Q[k]           for every k   (:261-264)
//               sum_k X[m,k] <= 1                                   for every m   (:267-268)
//               rate(r,j) >= rate(r,j+1)  where n_now[(r,j+1)] == 0               (:270-271)
//
// The reference hands this to docplex/CPLEX, which is absent here and whose
// optimum is not unique, so the solution x is treated as an
// INPUT of the accelerated path and this solver is its single source.  The
// upper bounds X <= 1 are implied by the machine rows and are not added.
#include "fjsp_host.h"

#include <cmath>
#include <cstddef>

namespace fjsp {

namespace {
// rowi[j] -= f * rowr[j]: the inner loop of a pivot (one rounding for the product, one for the difference:
// the file is compiled with -ffp-contract=off and the AVX2 clone is built WITHOUT the fma feature, so both
// clones perform the same IEEE operations element by element and produce identical bits).
#if defined(__HIP_DEVICE_COMPILE__)      // (hipcc also runs its device pass over this host-only file)
#define FJSP_HOST_CLONES
#else
#define FJSP_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif
FJSP_HOST_CLONES
void row_axpy(double *__restrict rowi, const double *__restrict rowr, double f, int n) {
    for (int j = 0; j < n; ++j) rowi[j] -= f * rowr[j];
}
FJSP_HOST_CLONES
void row_scale(double *__restrict row, double piv, int n) {
    for (int j = 0; j < n; ++j) row[j] /= piv;
}
constexpr double kEpsCost = 1e-9;   // entering threshold on reduced cost
constexpr double kEpsPiv = 1e-9;    // minimum pivot element
constexpr double kEpsZero = 1e-11;  // |x| below this is reported as exactly 0 (x != 0 test, :290)
}  // namespace

int solve_fluid_lp(int R, int M, const int *Jr, const int *p, const int *Q, const int *n_now,
                   double *x, double *objective) {
    std::vector<int> koff(R + 1, 0);
    for (int r = 0; r < R; ++r) koff[r + 1] = koff[r] + Jr[r];
    const int K = koff[R];
    // columns: eligible (m,k) pairs sorted by (m,k) -- the order the oracle shim
    // creates the docplex variables in -- then t.
    std::vector<int> col_of((size_t)K * M, -1);
    int nx = 0;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k)
            if (p[(size_t)k * M + m] > 0) col_of[(size_t)k * M + m] = nx++;
    const int tcol = nx;
    const int nv = nx + 1;
    // precedence rows
    std::vector<int> prec;  // k such that row rate(k+1) - rate(k) <= 0 exists
    for (int r = 0; r < R; ++r)
        for (int j = 0; j + 1 < Jr[r]; ++j)
            if (n_now[koff[r] + j + 1] == 0) prec.push_back(koff[r] + j);
    const int nr = K + M + (int)prec.size();
    const int nc = nv + nr + 1;
    const int rhs = nc - 1;
    std::vector<double> T((size_t)nr * nc, 0.0), z((size_t)nc, 0.0);
    std::vector<int> basis(nr);
    auto at = [&](int i, int j) -> double & { return T[(size_t)i * nc + j]; };
    for (int k = 0; k < K; ++k) {
        if (Q[k] <= 0) { set_error("fluid LP: Q[k] <= 0"); return -1; }
        bool any = false;
        for (int m = 0; m < M; ++m) {
            int c = col_of[(size_t)k * M + m];
            if (c < 0) continue;
            any = true;
            double rate = 1.0 / (double)p[(size_t)k * M + m];
            at(k, c) = -(rate / (double)Q[k]);
        }
        if (!any) { set_error("fluid LP: operation type without eligible machine"); return -1; }
        at(k, tcol) = 1.0;
    }
    for (int m = 0; m < M; ++m) {
        for (int k = 0; k < K; ++k) {
            int c = col_of[(size_t)k * M + m];
            if (c >= 0) at(K + m, c) = 1.0;
        }
        at(K + m, rhs) = 1.0;
    }
    for (size_t q = 0; q < prec.size(); ++q) {
        int k = prec[q], row = K + M + (int)q;
        for (int m = 0; m < M; ++m) {
            int c1 = col_of[(size_t)(k + 1) * M + m], c0 = col_of[(size_t)k * M + m];
            if (c1 >= 0) at(row, c1) += 1.0 / (double)p[(size_t)(k + 1) * M + m];
            if (c0 >= 0) at(row, c0) -= 1.0 / (double)p[(size_t)k * M + m];
        }
    }
    for (int i = 0; i < nr; ++i) { at(i, nv + i) = 1.0; basis[i] = nv + i; }
    z[tcol] = -1.0;  // maximise t

    const long max_iter = 200L * (nr + nc) + 1000;
    for (long it = 0;; ++it) {
        if (it > max_iter) { set_error("fluid LP: iteration limit"); return -1; }
        int s = -1;
        double best = -kEpsCost;
        for (int j = 0; j < nc - 1; ++j)
            if (z[j] < best) { best = z[j]; s = j; }
        if (s < 0) break;  // optimal
        // lexicographic ratio test
        int r = -1;
        for (int i = 0; i < nr; ++i) {
            double a = at(i, s);
            if (a <= kEpsPiv) continue;
            if (r < 0) { r = i; continue; }
            double ar = at(r, s);
            // compare rows i and r: (rhs, slack columns...) / pivot element
            double vi = at(i, rhs) / a, vr = at(r, rhs) / ar;
            double tol = 1e-12 * (std::fabs(vr) > 1.0 ? std::fabs(vr) : 1.0);
            if (vi < vr - tol) { r = i; continue; }
            if (vi > vr + tol) continue;
            for (int c = nv; c < nv + nr; ++c) {
                double wi = at(i, c) / a, wr = at(r, c) / ar;
                if (wi < wr) { r = i; break; }
                if (wi > wr) break;
            }
        }
        if (r < 0) { set_error("fluid LP: unbounded"); return -1; }
        // pivot
        double piv = at(r, s);
        double *rowr = &T[(size_t)r * nc];
        row_scale(rowr, piv, nc);
        rowr[s] = 1.0;
        for (int i = 0; i < nr; ++i) {
            if (i == r) continue;
            double f = at(i, s);
            if (f == 0.0) continue;
            double *rowi = &T[(size_t)i * nc];
            row_axpy(rowi, rowr, f, nc);
            rowi[s] = 0.0;
            if (rowi[rhs] < 0.0 && rowi[rhs] > -1e-12) rowi[rhs] = 0.0;
        }
        double f = z[s];
        if (f != 0.0) {
            row_axpy(z.data(), rowr, f, nc);
            z[s] = 0.0;
        }
        basis[r] = s;
    }
    for (size_t i = 0; i < (size_t)K * M; ++i) x[i] = 0.0;
    std::vector<double> val((size_t)nv, 0.0);
    for (int i = 0; i < nr; ++i)
        if (basis[i] < nv) val[basis[i]] = at(i, rhs);
    for (int k = 0; k < K; ++k)
        for (int m = 0; m < M; ++m) {
            int c = col_of[(size_t)k * M + m];
            if (c < 0) continue;
            double v = val[c];
            if (v < kEpsZero) v = 0.0;
            if (v > 1.0) v = 1.0;
            x[(size_t)k * M + m] = v;
        }
    if (objective) *objective = val[tcol];
    // every operation type must keep a positive fluid rate (fluid_time_sum = 1/rate_sum, :295)
    for (int k = 0; k < K; ++k) {
        double sacc = 0.0;
        for (int m = 0; m < M; ++m)
            if (p[(size_t)k * M + m] > 0) sacc += x[(size_t)k * M + m] / (double)p[(size_t)k * M + m];
        if (!(sacc > 0.0)) { set_error("fluid LP: zero rate for an operation type"); return -1; }
    }
    return 0;
}

}  // namespace fjsp
