// The fluid-model LP of an order arrival (environments/class_FJSSP.py:246-280, class_MODFJSP.py:240-280) ON THE DEVICE:
// one workgroup per parked environment, the simplex tableau in LDS.
//
// csrc/fjsp_lp.cpp is the product's LP solver (dense primal simplex, Dantzig pricing with the first smallest reduced cost,
// lexicographic ratio test) and its solution x is an input of the environment kernels; at an order arrival the blocking
// service of fjsp_env.hip used to bring the LP inputs to the host, solve there and upload x -- one stream synchronisation
// and 0.05-0.8 ms of host time per vector step.  This file restates that solver pivot for pivot for the GPU:
//   * same tableau (rows: one per operation type, one per machine, one per precedence constraint; columns: eligible
//     (m, k) pairs in (m, k) order, t, the slacks, the right-hand side), same entering rule, same ratio test -- a strictly
//     sequential scan over the rows with its tolerances, run by one lane over values the other lanes have laid out --, same
//     elimination arithmetic (f64 divide, multiply, subtract: -ffp-contract=off, no FMA; the host's AVX clones are built
//     without FMA for the same reason), same clean-up of tiny negative right-hand sides, same extraction of x;
//   * so x is BIT-IDENTICAL to fjsp_lp.cpp's (tests/test_gpu_parity.py::test_device_lp_equals_the_host_lp), and the
//     environment's trajectory does not depend on where its LPs were solved;
//   * the tableau of the reference's industrial instances (K = 31, M = 20: 79 rows x 137 columns) is 87 KB: it lives in
//     LDS (160 KB per CU); a batch whose largest possible tableau does not fit keeps the host service (fjsp_env.hip
//     decides at create time: lp_device_lds_bytes);
//   * the launch needs no host round trip: the workgroups read the number of parked environments from the pending list
//     the step kernel filled (DevBatch::pending_count) and stride over the slots.
#include <hip/hip_runtime.h>

#include "../../include/fjsp_amd.h"
#include "fjsp_common.h"
#include "fjsp_device.h"

#pragma clang fp contract(off)

namespace fjsp {

namespace {
constexpr double kEpsCost = 1e-9;   // entering threshold on reduced cost            (fjsp_lp.cpp)
constexpr double kEpsPiv = 1e-9;    // minimum pivot element
constexpr double kEpsZero = 1e-11;  // |x| below this is reported as exactly 0 (x != 0 test, class_FJSSP.py:290)
constexpr int kThreads = 512;

extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];

struct LpDims { int K, M, nx, nv, nr, nc, nprec; };
}  // namespace

// LDS bytes of the largest tableau an instance of (K operation types, M machines, nx eligible pairs, R kinds) can need
size_t lp_device_lds_bytes(int K, int M, int nx, int R, int MP) {
    const size_t nr = (size_t)K + M + (K - R), nc = (size_t)nx + 1 + nr + 1;
    // tableau | column values (x extraction) | (spare) | basis | ... | col_of, prec list | staged inputs: p, Q, n_now, kB (the tail)
    const size_t bytes = nr * nc * 8 + nc * 8 + 2 * nr * 8 + nr * 4 + (size_t)K * M * 2 + (size_t)K * 2 + nr * 2 + (size_t)K * MP * 2 + (size_t)K * 8 + 128;
    return (bytes + 15) & ~(size_t)15;
}

namespace {
constexpr int kZT = 8;              // registers of a lane for the objective row / the pivot row: columns l, l + 64, ... (nc <= 512)

__device__ inline double lane_f64(double v, int lane) {     // v of a wave-uniform lane
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
}  // namespace

namespace {
template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
}  // namespace

namespace {
__device__ inline double wave_min_f64(double x) {                  // the smallest x of the wave, in every lane ("<": NaNs are passed over)
#define LP_MIN1(CTRL) { const double o = dpp_f64<CTRL>(x); if (o < x) x = o; }
    LP_MIN1(0xB1) LP_MIN1(0x4E) LP_MIN1(0x141) LP_MIN1(0x140)
#undef LP_MIN1
    double m = lane_f64(x, 0);
#pragma unroll
    for (int q = 16; q < 64; q += 16) { const double o = lane_f64(x, q); if (o < m) m = o; }
    return m;
}
}  // namespace

int lp_device_max_columns() { return kZT * 64; }

namespace {
struct LpTab { double *T; int *basis; int nr, nc, nv, tcol; };

__device__ inline double wave_fmin_f64(double x) {                 // the smallest x of the wave (no NaNs among them), in every lane
    x = __builtin_fmin(x, dpp_f64<0xB1>(x));
    x = __builtin_fmin(x, dpp_f64<0x4E>(x));
    x = __builtin_fmin(x, dpp_f64<0x141>(x));
    x = __builtin_fmin(x, dpp_f64<0x140>(x));
    return __builtin_fmin(__builtin_fmin(lane_f64(x, 0), lane_f64(x, 16)), __builtin_fmin(lane_f64(x, 32), lane_f64(x, 48)));
}

__device__ inline uint32_t wave_min_u32(uint32_t x) {              // the smallest x of the wave, in every lane
#define LP_UMIN(CTRL) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false); x = o < x ? o : x; }
    LP_UMIN(0xB1) LP_UMIN(0x4E) LP_UMIN(0x141) LP_UMIN(0x140)
#undef LP_UMIN
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
constexpr int kLexCols = 16;        // slack columns of a tie-break step: their signs fit one 32-bit signature

// The pivots of one LP (every thread of the workgroup; returns the failure code, 0 = optimal).
//
// A pivot costs two workgroup barriers.  Every wave holds the objective row in registers and chooses the entering column
// and the leaving row BY ITSELF (the same values, the same operations: the same answer in every wave, nothing to
// exchange).  Rows of the elimination are dealt to the waves; the scaled pivot row travels in registers.  The waves of
// a workgroup share four SIMDs: a single wave issues a dependent instruction every ~10 cycles, the others fill the gaps.
template <int NT>
__device__ __forceinline__ int lp_pivots(const LpTab tab, const int w, const int l, const int tid, long &n_piv) {
    double *const T = tab.T;
    const int nr = tab.nr, nc = tab.nc, nv = tab.nv, rhs = nc - 1, tcol = tab.tcol;
    constexpr int RB = NT <= 4 ? 4 : 2;                                // rows of the elimination a wave has in flight
    constexpr int kWaves = kThreads / 64;
    auto at = [&](int i, int j) -> double & { return T[(size_t)i * nc + j]; };
    const double inf = __builtin_huge_val();
    double zr[NT];                                                     // the objective row, in every wave: maximise t
#pragma unroll
    for (int t = 0; t < NT; ++t) zr[t] = (l + 64 * t == tcol) ? -1.0 : 0.0;
    const long max_iter = 200L * (nr + nc) + 1000;
    for (long it = 0;; ++it) {
        if (it > max_iter) return 2;                                   // "iteration limit"
        // ---- entering column: the first smallest reduced cost below -eps (the smallest value, then its first column)
        int s = -1;
        {
            double m = inf;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (l + 64 * t < nc - 1) m = __builtin_fmin(m, zr[t]);
            m = wave_fmin_f64(m);
            if (!(m < -kEpsCost)) return 0;                            // optimal
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned long long hit = __ballot(l + 64 * t < nc - 1 && zr[t] == m);
                if (s < 0 && hit) s = 64 * t + __builtin_ctzll(hit);
            }
        }
        // ---- lexicographic ratio test (fjsp_lp.cpp): a strictly sequential scan over the rows with a > eps, which compares
        // the row it meets with the best so far -- by the ratio v, and inside a tolerance by the slack columns over the
        // pivot element, lexicographically.  These LPs are degenerate (every operation row and precedence row has a zero
        // right-hand side): most rows tie, and pairwise tie-breaks would be where a pivot's time goes.  When the rows
        // split cleanly into those with exactly the smallest ratio and those the scan's own two tests (evaluated here
        // with its expressions) put strictly beyond the tolerance from them, the scan's result is the FIRST
        // LEXICOGRAPHIC MINIMUM among the former -- an order-independent quantity -- and the whole set is narrowed
        // column by column with the rows in lanes; anything else (near-ties with different ratios, more than 128 rows)
        // takes the sequential scan below.
        int r = -1;
        double ar = 0.0, vr = 0.0;
        double acol[2] = {0.0, 0.0};          // column s of rows l and 64 + l: the elimination's factors (later rows: from LDS)
        bool chosen = false;
        if (nr <= 128) {
            const int i0 = l, i1 = 64 + l;
            const double a0 = i0 < nr ? at(i0, s) : 0.0, a1 = i1 < nr ? at(i1, s) : 0.0;
            acol[0] = a0; acol[1] = a1;
            const bool el0 = a0 > kEpsPiv, el1 = a1 > kEpsPiv;
            const double v0 = el0 ? at(i0, rhs) / a0 : 0.0, v1 = el1 ? at(i1, rhs) / a1 : 0.0;
            if (!(__ballot(el0) | __ballot(el1))) return 3;            // "unbounded"
            double x = el0 ? v0 : inf;
            if (el1 && v1 < x) x = v1;
            const double vmin = wave_fmin_f64(x);
            const double tolmin = 1e-12 * (fabs(vmin) > 1.0 ? fabs(vmin) : 1.0), hi = vmin + tolmin;
            const bool in0 = el0 && v0 == vmin, in1 = el1 && v1 == vmin;
            const double tol0 = 1e-12 * (fabs(v0) > 1.0 ? fabs(v0) : 1.0), tol1 = 1e-12 * (fabs(v1) > 1.0 ? fabs(v1) : 1.0);
            const bool far0 = v0 > hi && vmin < v0 - tol0, far1 = v1 > hi && vmin < v1 - tol1;
            const unsigned long long bad = __ballot(el0 && !in0 && !far0) | __ballot(el1 && !in1 && !far1);
            if (!bad) {
                bool k0 = in0, k1 = in1;                               // the rows still in the race
                int cnt = __builtin_popcountll(__ballot(k0)) + __builtin_popcountll(__ballot(k1));
                const int cend = nv + nr;
                for (int c = nv; c < cend && cnt > 1; c += kLexCols) {
                    double t0[kLexCols], t1[kLexCols];
                    const bool any1 = __ballot(k1) != 0ull;
                    bool small = false;                                // a nonzero entry whose quotient could underflow
                    // sign signatures of the next 16 columns, first column in the top bits: negative 0 < zero 1 < positive 2
                    // (x / a keeps x's sign and is nonzero: a > 1e-9 and |x| >= 1e-280) -- rows order by them as by their
                    // quotients wherever the signs differ
                    uint32_t sig0 = 0u, sig1 = 0u;
#pragma unroll
                    for (int u = 0; u < kLexCols; ++u) {
                        t0[u] = (k0 && c + u < cend) ? at(i0, c + u) : 0.0;
                        t1[u] = (any1 && k1 && c + u < cend) ? at(i1, c + u) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < kLexCols; ++u) {
                        small = small || (t0[u] != 0.0 && !(fabs(t0[u]) >= 1e-280)) || (t1[u] != 0.0 && !(fabs(t1[u]) >= 1e-280));
                        sig0 = (sig0 << 2) | (uint32_t)((t0[u] == 0.0 ? 1 : 0) + (t0[u] > 0.0 ? 2 : 0));
                        sig1 = (sig1 << 2) | (uint32_t)((t1[u] == 0.0 ? 1 : 0) + (t1[u] > 0.0 ? 2 : 0));
                    }
                    const bool exact_signs = __ballot(small) == 0ull;
                    if (exact_signs) {
                        if (!k0) sig0 = 0xFFFFFFFFu;
                        if (!k1) sig1 = 0xFFFFFFFFu;
                        const uint32_t smin = wave_min_u32(sig0 < sig1 ? sig0 : sig1);
                        const uint32_t d = smin ^ 0x55555555u;                         // 0: the best rows are zero in all 16 columns
                        // the rows that agree with the best signature up to and including its first nonzero sign
                        const uint32_t keep = d ? ~((1u << (2 * ((31 - __builtin_clz(d)) >> 1))) - 1u) : 0xFFFFFFFFu;
                        const bool p0 = k0 && ((sig0 ^ smin) & keep) == 0u, p1 = k1 && ((sig1 ^ smin) & keep) == 0u;
                        const int np = __builtin_popcountll(__ballot(p0)) + __builtin_popcountll(__ballot(p1));
                        if (d == 0u || np == 1) { k0 = p0; k1 = p1; cnt = np; continue; }
                        // (several rows share a nonzero sign in the deciding column: magnitudes decide, column by column below)
                    }
#pragma unroll
                    for (int u = 0; u < kLexCols; ++u) {
                        if (c + u >= cend || cnt <= 1) break;
                        const double x0 = t0[u], x1 = t1[u];
                        bool s0 = k0, s1 = k1;                         // the rows whose quotients are compared
                        // signs decide most columns without a division: one negative entry wins, positives lose against zeros
                        if (exact_signs) {
                            const bool g0 = k0 && x0 < 0.0, g1 = k1 && x1 < 0.0;
                            const int nn = __builtin_popcountll(__ballot(g0)) + (any1 ? __builtin_popcountll(__ballot(g1)) : 0);
                            if (nn == 1) { k0 = g0; k1 = g1; cnt = 1; continue; }
                            if (nn == 0) {
                                const bool z0 = k0 && x0 == 0.0, z1 = k1 && x1 == 0.0;
                                const int nz = __builtin_popcountll(__ballot(z0)) + (any1 ? __builtin_popcountll(__ballot(z1)) : 0);
                                if (nz > 0) { k0 = z0; k1 = z1; cnt = nz; continue; }
                            } else { s0 = g0; s1 = g1; }
                        }
                        const double w0 = s0 ? x0 / a0 : inf, w1 = s1 ? x1 / a1 : inf;
                        const double wm = wave_fmin_f64(__builtin_fmin(w0, w1));
                        k0 = s0 && w0 == wm; k1 = s1 && w1 == wm;
                        cnt = __builtin_popcountll(__ballot(k0)) + __builtin_popcountll(__ballot(k1));
                    }
                }
                const unsigned long long E0 = __ballot(k0), E1 = __ballot(k1);
                if (E0 | E1) {
                    chosen = true;
                    vr = vmin;
                    if (E0) { const int q = __builtin_ctzll(E0); r = q; ar = lane_f64(a0, q); }
                    else { const int q = __builtin_ctzll(E1); r = 64 + q; ar = lane_f64(a1, q); }
                }
            }
        }
        if (!chosen)
            for (int base = 0; base < nr; base += 64) {
                const int i = base + l;
                const double a = i < nr ? at(i, s) : 0.0;
                if (base == 0) acol[0] = a;
                if (base == 64) acol[1] = a;
                const bool el = a > kEpsPiv;
                const double v = el ? at(i, rhs) / a : 0.0;
                unsigned long long mask = __ballot(el);
                while (mask) {
                    // rows the scan would pass with "vi > vr + tol: continue" are passed in one step: the next row it
                    // looks at closer is the first remaining one for which that test fails
                    if (r >= 0) {
                        const double tol = 1e-12 * (fabs(vr) > 1.0 ? fabs(vr) : 1.0);
                        mask &= __ballot(el && !(v > vr + tol));
                        if (!mask) break;
                    }
                    const int q = __builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    const int irow = base + q;
                    const double ai = lane_f64(a, q), vi = lane_f64(v, q);
                    if (r < 0) { r = irow; ar = ai; vr = vi; continue; }
                    const double tol = 1e-12 * (fabs(vr) > 1.0 ? fabs(vr) : 1.0);
                    if (vi < vr - tol) { r = irow; ar = ai; vr = vi; continue; }
                    for (int c0 = nv; c0 < nv + nr; c0 += 64) {         // a tie: 64 slack columns at a time
                        const int c = c0 + l;
                        const bool in = c < nv + nr;
                        const double wi = in ? at(irow, c) / ai : 0.0, wr = in ? at(r, c) / ar : 0.0;
                        const unsigned long long lt = __ballot(in && wi < wr), gt = __ballot(in && wi > wr);
                        if (lt | gt) {
                            const int f = __builtin_ctzll(lt | gt);
                            if ((lt >> f) & 1ull) { r = irow; ar = ai; vr = vi; }
                            break;
                        }
                    }
                }
            }
        if (r < 0) return 3;                                           // "unbounded"
        // ---- pivot: the scaled row r into registers (column s becomes exactly 1), the objective's factor from its lane
        const double piv = ar;
        const double *rowr = &T[(size_t)r * nc];
        double rr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = l + 64 * t;
            rr[t] = j < nc ? (j == s ? 1.0 : rowr[j] / piv) : 0.0;
        }
        double fz = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if ((s >> 6) == t) fz = lane_f64(zr[t], s & 63);
        __syncthreads();                      // every wave has read column s, the right-hand sides and row r
        // ---- elimination.  Column s is sparse: only rows with a nonzero factor change.  Those rows (a ballot of the
        // factors every wave holds) are dealt round-robin to the waves, and a wave takes RB of its rows at a time -- all
        // their loads in flight together; lane l has the columns l, l + 64, ...  (x - f * 1 of column s is exactly 0, as
        // the host writes it; the clean-up of a tiny negative right-hand side touches one lane of one chunk.)
        const int rhs_t = rhs >> 6, rhs_l = rhs & 63;
        {
            const int i0 = l, i1 = 64 + l;
            const bool nz0 = i0 < nr && i0 != r && acol[0] != 0.0, nz1 = i1 < nr && i1 != r && acol[1] != 0.0;
            const unsigned long long Z0 = __ballot(nz0), Z1 = __ballot(nz1), below = (1ull << l) - 1ull;
            const int rank0 = __builtin_popcountll(Z0 & below), rank1 = __builtin_popcountll(Z0) + __builtin_popcountll(Z1 & below);
            unsigned long long my0 = __ballot(nz0 && (rank0 & (kWaves - 1)) == w);
            unsigned long long my1 = __ballot(nz1 && (rank1 & (kWaves - 1)) == w);
            while (my0 | my1) {
                int row[RB];
                double f[RB], x[RB][NT];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (my0) { const int q = __builtin_ctzll(my0); my0 &= my0 - 1ull; row[u] = q; f[u] = lane_f64(acol[0], q); }
                    else if (my1) { const int q = __builtin_ctzll(my1); my1 &= my1 - 1ull; row[u] = 64 + q; f[u] = lane_f64(acol[1], q); }
                    else { row[u] = -1; f[u] = 0.0; }
                }
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (row[u] >= 0) {
                        const double *rowi = &T[(size_t)row[u] * nc];
#pragma unroll
                        for (int t = 0; t < NT; ++t) { const int j = l + 64 * t; x[u][t] = j < nc ? rowi[j] : 0.0; }
                    }
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (row[u] >= 0) {
                        double *rowi = &T[(size_t)row[u] * nc];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int j = l + 64 * t;
                            double v = x[u][t] - f[u] * rr[t];
                            if (t == rhs_t && l == rhs_l && v < 0.0 && v > -1e-12) v = 0.0;
                            if (j < nc) rowi[j] = v;
                        }
                    }
            }
            if (w == kWaves - 1) {                // the scaled pivot row (nobody reads row r between the two barriers)
                double *rowi = &T[(size_t)r * nc];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = l + 64 * t;
                    if (j < nc) rowi[j] = rr[t];
                }
            }
        }
        for (int i = 128 + w; i < nr; i += kWaves) {                  // (rows beyond the factors held in lanes)
            if (i == r) continue;
            double *rowi = &T[(size_t)i * nc];
            const double f = rowi[s];
            if (f == 0.0) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = l + 64 * t;
                if (j < nc) {
                    double v = rowi[j] - f * rr[t];
                    if (j == rhs && v < 0.0 && v > -1e-12) v = 0.0;
                    rowi[j] = v;
                }
            }
        }
        if (fz != 0.0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) zr[t] = zr[t] - fz * rr[t];
        }
        if (tid == 0) tab.basis[r] = s;
        ++n_piv;
        __syncthreads();
    }
}
}  // namespace

// One workgroup per parked environment (slot): solves its LP, writes x to lp_x[slot] (f64[KP][MP], zeros elsewhere).
// err[0] becomes nonzero when an LP fails (infeasible input, unbounded, iteration limit): the host reports it at the next
// synchronising call.
// (the set-up here, the pivots in lp_pivots above, then x out of the final basis)
__global__ __launch_bounds__(kThreads) void lp_device_kernel(DevBatch b, const uint32_t *count_dev, int count_host, const uint32_t *ids,
                                                             const uint16_t *lp_in, double *lp_x, uint32_t *err, unsigned long long *solved,
                                                             uint32_t lds_bytes) {
    const int tid = (int)threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const uint32_t count = count_dev ? min(*count_dev, (uint32_t)b.N) : (uint32_t)count_host;
    if (blockIdx.x == 0 && tid == 0 && solved) atomicAdd(solved, (unsigned long long)count);
    for (uint32_t slot = blockIdx.x; slot < count; slot += gridDim.x) {
        const int env = (int)ids[slot];
        const int inst = b.n_inst == b.N ? env : env % b.n_inst;
        const unsigned char *ir = b.inst + (size_t)inst * b.L.i_stride;
        const InstHeader h = *reinterpret_cast<const InstHeader *>(ir);
        const int K = h.K, M = h.M, MP = b.MP, KP = b.KP;
        const uint16_t *p_g = reinterpret_cast<const uint16_t *>(ir + b.L.i_p);         // [KP][MP], 0 = ineligible
        const uint32_t *kB_g = reinterpret_cast<const uint32_t *>(ir + b.L.i_kB);
        const uint16_t *Q_g = lp_in + (size_t)slot * 2 * KP;
        double *xout = lp_x + (size_t)slot * KP * MP;
        // the inputs once into LDS (coalesced): everything below reads them there
        // (the tail of the allocation: its place does not depend on the tableau's size)
        unsigned char *tail = lp_lds + lds_bytes;
        uint16_t *p = reinterpret_cast<uint16_t *>(tail - (size_t)K * MP * 2 - (size_t)K * 8 - 16);     // [K][MP]
        uint16_t *Q = p + (size_t)K * MP, *now = Q + K;
        uint32_t *kB = reinterpret_cast<uint32_t *>(tail - (size_t)K * 4 - 8);
        __shared__ LpDims dims;
        __shared__ int s_fail;
        __syncthreads();                                                                   // (the previous slot's readers are done)
        for (int q = tid; q < K * MP; q += kThreads) p[q] = p_g[q];
        for (int q = tid; q < K; q += kThreads) { Q[q] = Q_g[q]; now[q] = Q_g[KP + q]; kB[q] = kB_g[q]; }
        if (tid == 0) s_fail = 0;
        __syncthreads();
        // ---- dimensions: columns = eligible pairs in (m, k) order, then t; precedence rows in k order (fjsp_lp.cpp).
        // The column numbers and the precedence list are prefix counts: wave 0 takes them 64 at a time from ballots.
        uint16_t *col_of = p - ((size_t)K * M + K + 8);                                    // [K][M] -> column, 0xFFFF = ineligible
        uint16_t *prec = col_of + (size_t)K * M;
        if (w == 0) {
            int nx = 0, nprec = 0;
            for (int base = 0; base < K * M; base += 64) {
                const int q = base + l, m = q / K, k = q - m * K;                          // (m, k) order
                const bool el = q < K * M && p[k * MP + m] > 0;
                const unsigned long long mask = __ballot(el);
                if (q < K * M) col_of[k * M + m] = el ? (uint16_t)(nx + __builtin_popcountll(mask & ((1ull << l) - 1ull))) : (uint16_t)0xFFFFu;
                nx += __builtin_popcountll(mask);
            }
            for (int base = 0; base + 1 < K; base += 64) {
                const int k = base + l;
                bool pr = false;
                if (k + 1 < K) {
                    const uint32_t kb = kB[k];
                    pr = (kb & 0xFFu) + 1u < ((kb >> 8) & 0xFFu) && now[k + 1] == 0;        // j + 1 < J_r: k + 1 is the same kind's next stage
                }
                const unsigned long long mask = __ballot(pr);
                if (pr) prec[nprec + __builtin_popcountll(mask & ((1ull << l) - 1ull))] = (uint16_t)k;
                nprec += __builtin_popcountll(mask);
            }
            if (l == 0) {
                dims.K = K; dims.M = M; dims.nx = nx; dims.nv = nx + 1; dims.nprec = nprec;
                dims.nr = K + M + nprec; dims.nc = nx + 1 + dims.nr + 1;
            }
        }
        __syncthreads();
        const int nv = dims.nv, nr = dims.nr, nc = dims.nc, tcol = dims.nx, rhs = nc - 1, nprec = dims.nprec;
        double *T = reinterpret_cast<double *>(lp_lds);
        double *val = T + (size_t)nr * nc;                                 // [nc]: value of a column's basic variable (x extraction)
        int *basis = reinterpret_cast<int *>(val + nc + 2 * (size_t)nr);
        auto at = [&](int i, int j) -> double & { return T[(size_t)i * nc + j]; };
        for (int q = tid; q < nr * nc; q += kThreads) T[q] = 0.0;
        __syncthreads();
        // ---- fill
        for (int k = tid; k < K; k += kThreads) {
            if (Q[k] == 0) s_fail = 1;                                                     // "fluid LP: Q[k] <= 0"
            bool any = false;
            for (int m = 0; m < M; ++m) {
                const uint16_t c = col_of[k * M + m];
                if (c == 0xFFFFu) continue;
                any = true;
                const double rate = 1.0 / (double)p[k * MP + m];
                at(k, c) = -(rate / (double)Q[k]);
            }
            if (!any) s_fail = 1;                                                          // "operation type without eligible machine"
            at(k, tcol) = 1.0;
        }
        for (int m = tid; m < M; m += kThreads) {
            for (int k = 0; k < K; ++k) {
                const uint16_t c = col_of[k * M + m];
                if (c != 0xFFFFu) at(K + m, c) = 1.0;
            }
            at(K + m, rhs) = 1.0;
        }
        for (int q = tid; q < nprec; q += kThreads) {
            const int k = prec[q], row = K + M + q;
            for (int m = 0; m < M; ++m) {
                const uint16_t c1 = col_of[(k + 1) * M + m], c0 = col_of[k * M + m];
                if (c1 != 0xFFFFu) at(row, c1) += 1.0 / (double)p[(k + 1) * MP + m];
                if (c0 != 0xFFFFu) at(row, c0) -= 1.0 / (double)p[k * MP + m];
            }
        }
        for (int i = tid; i < nr; i += kThreads) { at(i, nv + i) = 1.0; basis[i] = nv + i; }
        __syncthreads();
        // ---- pivots (the loop is compiled for 2, 3, 4, 6 and 8 chunks of 64 columns: the rows a lane handles stay in registers)
        LpTab tab{T, basis, nr, nc, nv, tcol};
        long n_piv = 0;
        int fail = s_fail;
        if (!fail) {
            const int nt = (nc + 63) >> 6;
            fail = nt <= 2 ? lp_pivots<2>(tab, w, l, tid, n_piv) : nt <= 3 ? lp_pivots<3>(tab, w, l, tid, n_piv)
                 : nt <= 4 ? lp_pivots<4>(tab, w, l, tid, n_piv) : nt <= 6 ? lp_pivots<6>(tab, w, l, tid, n_piv)
                                                                           : lp_pivots<8>(tab, w, l, tid, n_piv);
        }
        __syncthreads();
        // ---- x out of the basis (values below 1e-11 are exact zeros, above 1 clamp to 1)
        for (int q = tid; q < KP * MP; q += kThreads) xout[q] = 0.0;
        for (int q = tid; q < nv; q += kThreads) val[q] = 0.0;
        __syncthreads();
        if (!fail) {
            for (int i = tid; i < nr; i += kThreads)
                if (basis[i] < nv) val[basis[i]] = at(i, rhs);
            __syncthreads();
            for (int q = tid; q < K * M; q += kThreads) {
                const int k = q / M, m = q % M;
                const uint16_t c = col_of[q];
                if (c == 0xFFFFu) continue;
                double v = val[c];
                if (v < kEpsZero) v = 0.0;
                if (v > 1.0) v = 1.0;
                xout[k * MP + m] = v;
            }
            __syncthreads();
            // every operation type must keep a positive fluid rate (fluid_time_sum = 1 / rate_sum, :295)
            for (int k = tid; k < K; k += kThreads) {
                double sacc = 0.0;
                for (int m = 0; m < M; ++m)
                    if (p[k * MP + m] > 0) sacc += xout[k * MP + m] / (double)p[k * MP + m];
                if (!(sacc > 0.0)) s_fail = 4;
            }
        }
        __syncthreads();
        if (tid == 0 && (fail || s_fail)) atomicOr(err, (uint32_t)(fail ? fail : s_fail));
        if (tid == 0 && solved) atomicAdd(solved + 1, (unsigned long long)n_piv);
    }
}

int launch_lp_device(const DevBatch &b, const uint32_t *count_dev, int count_host, const uint32_t *ids, const uint16_t *lp_in, double *lp_x,
                     uint32_t *err, unsigned long long *solved, size_t lds, hipStream_t st) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lp_device_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = count_dev ? (b.N < 256 ? b.N : 256) : count_host;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(lp_device_kernel, dim3((unsigned)grid), dim3(kThreads), lds, st, b, count_dev, count_host, ids, lp_in, lp_x, err, solved, (uint32_t)lds);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace fjsp
