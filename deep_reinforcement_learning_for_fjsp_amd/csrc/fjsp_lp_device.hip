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
constexpr int kThreads = 256;

extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];

struct LpDims { int K, M, nx, nv, nr, nc, nprec; };
}  // namespace

// LDS bytes of the largest tableau an instance of (K operation types, M machines, nx eligible pairs, R kinds) can need
size_t lp_device_lds_bytes(int K, int M, int nx, int R, int MP) {
    const size_t nr = (size_t)K + M + (K - R), nc = (size_t)nx + 1 + nr + 1;
    // tableau | z | column a, column v (ratio test) | basis | col_of | prec list | eligible rows | staged inputs: p, Q, n_now, kB
    const size_t bytes = nr * nc * 8 + nc * 8 + 2 * nr * 8 + nr * 4 + (size_t)K * M * 2 + (size_t)K * 2 + nr * 2 + (size_t)K * MP * 2 + (size_t)K * 8 + 128;
    return (bytes + 15) & ~(size_t)15;
}

// One workgroup per parked environment (slot): solves its LP, writes x to lp_x[slot] (f64[KP][MP], zeros elsewhere).
// err[0] becomes nonzero when an LP fails (infeasible input, unbounded, iteration limit): the host reports it at the next
// synchronising call.
__global__ __launch_bounds__(kThreads) void lp_device_kernel(DevBatch b, const uint32_t *count_dev, int count_host, const uint32_t *ids,
                                                             const uint16_t *lp_in, double *lp_x, uint32_t *err, unsigned long long *solved,
                                                             uint32_t lds_bytes) {
    const int tid = (int)threadIdx.x;
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
        // the inputs once into LDS (coalesced): everything below -- one lane's sequential scans included -- reads them there
        // (the tail of the allocation: its place does not depend on the tableau's size)
        unsigned char *tail = lp_lds + lds_bytes;
        uint16_t *p = reinterpret_cast<uint16_t *>(tail - (size_t)K * MP * 2 - (size_t)K * 8 - 16);     // [K][MP]
        uint16_t *Q = p + (size_t)K * MP, *now = Q + K;
        uint32_t *kB = reinterpret_cast<uint32_t *>(tail - (size_t)K * 4 - 8);
        for (int q = tid; q < K * MP; q += kThreads) p[q] = p_g[q];
        for (int q = tid; q < K; q += kThreads) { Q[q] = Q_g[q]; now[q] = Q_g[KP + q]; kB[q] = kB_g[q]; }
        __syncthreads();
        // ---- dimensions: columns = eligible pairs in (m, k) order, then t; precedence rows in k order (fjsp_lp.cpp)
        __shared__ LpDims dims;
        __shared__ int s_enter, s_leave, s_fail, s_nel;
        __shared__ double s_red[kThreads / 64];
        __shared__ int s_redi[kThreads / 64];
        // carve (sizes depend on the instance; offsets computed by every thread alike)
        int nx = 0, nprec = 0;
        if (tid == 0) {
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < K; ++k) nx += p[k * MP + m] > 0 ? 1 : 0;
            for (int k = 0; k + 1 < K; ++k) {
                const uint32_t kb = kB[k];
                const bool has_next = (kb & 0xFFu) + 1u < ((kb >> 8) & 0xFFu);                  // j + 1 < J_r: k + 1 is the same kind's next stage
                if (has_next && now[k + 1] == 0) nprec++;
            }
            dims.K = K; dims.M = M; dims.nx = nx; dims.nv = nx + 1; dims.nprec = nprec;
            dims.nr = K + M + nprec; dims.nc = nx + 1 + dims.nr + 1;
            s_fail = 0;
        }
        __syncthreads();
        const int nv = dims.nv, nr = dims.nr, nc = dims.nc, tcol = dims.nx, rhs = nc - 1;
        double *T = reinterpret_cast<double *>(lp_lds);
        double *z = T + (size_t)nr * nc;
        double *cola = z + nc;             // column s of the tableau (ratio test, elimination factors)
        double *colv = cola + nr;          // rhs / column s
        int *basis = reinterpret_cast<int *>(colv + nr);
        uint16_t *col_of = reinterpret_cast<uint16_t *>(basis + nr);       // [K][M] -> column, 0xFFFF = ineligible
        uint16_t *prec = col_of + (size_t)K * M;
        uint16_t *elig_rows = prec + K;                                    // rows with a pivot candidate in the entering column
        auto at = [&](int i, int j) -> double & { return T[(size_t)i * nc + j]; };
        for (int q = tid; q < nr * nc; q += kThreads) T[q] = 0.0;
        for (int q = tid; q < nc; q += kThreads) z[q] = 0.0;
        if (tid == 0) {
            int c = 0;
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < K; ++k) col_of[k * M + m] = p[k * MP + m] > 0 ? (uint16_t)c++ : (uint16_t)0xFFFFu;
            int q = 0;
            for (int k = 0; k + 1 < K; ++k) {
                const uint32_t kb = kB[k];
                if ((kb & 0xFFu) + 1u < ((kb >> 8) & 0xFFu) && now[k + 1] == 0) prec[q++] = (uint16_t)k;
            }
        }
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
        for (int q = tid; q < dims.nprec; q += kThreads) {
            const int k = prec[q], row = K + M + q;
            for (int m = 0; m < M; ++m) {
                const uint16_t c1 = col_of[(k + 1) * M + m], c0 = col_of[k * M + m];
                if (c1 != 0xFFFFu) at(row, c1) += 1.0 / (double)p[(k + 1) * MP + m];
                if (c0 != 0xFFFFu) at(row, c0) -= 1.0 / (double)p[k * MP + m];
            }
        }
        for (int i = tid; i < nr; i += kThreads) { at(i, nv + i) = 1.0; basis[i] = nv + i; }
        if (tid == 0) z[tcol] = -1.0;        // maximise t
        __syncthreads();
        // ---- pivots
        const long max_iter = 200L * (nr + nc) + 1000;
        for (long it = 0; !s_fail; ++it) {
            if (it > max_iter) { if (tid == 0) s_fail = 2; break; }                        // "iteration limit"
            // entering column: the first smallest reduced cost below -eps
            double best = -kEpsCost;
            int s = -1;
            for (int j = tid; j < nc - 1; j += kThreads)
                if (z[j] < best) { best = z[j]; s = j; }
            // (a thread visits its columns in increasing order: it holds its first minimum; reduce by (value, index))
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_down(best, off, 64);
                const int os = __shfl_down(s, off, 64);
                if (os >= 0 && (s < 0 || ob < best || (ob == best && os < s))) { best = ob; s = os; }
            }
            if ((tid & 63) == 0) { s_red[tid >> 6] = best; s_redi[tid >> 6] = s; }
            __syncthreads();
            if (tid == 0) {
                double bb = s_red[0]; int bs = s_redi[0];
                for (int w = 1; w < kThreads / 64; ++w) {
                    const double ob = s_red[w]; const int os = s_redi[w];
                    if (os >= 0 && (bs < 0 || ob < bb || (ob == bb && os < bs))) { bb = ob; bs = os; }
                }
                s_enter = bs;
            }
            __syncthreads();
            s = s_enter;
            if (s < 0) break;                 // optimal
            // column s and the ratios, laid out for the sequential ratio test
            for (int i = tid; i < nr; i += kThreads) {
                const double a = at(i, s);
                cola[i] = a;
                colv[i] = a > kEpsPiv ? at(i, rhs) / a : 0.0;
            }
            __syncthreads();
            if (tid < 64) {
                // the rows the scan below looks at (a > eps), in row order: ballots of wave 0 over blocks of 64 rows
                int n_el = 0;
                for (int base = 0; base < nr; base += 64) {
                    const int i = base + tid;
                    const bool el = i < nr && cola[i] > kEpsPiv;
                    const unsigned long long m = __ballot(el);
                    if (el) elig_rows[n_el + __builtin_popcountll(m & ((1ull << tid) - 1ull))] = (uint16_t)i;
                    n_el += __builtin_popcountll(m);
                }
                if (tid == 0) s_nel = n_el;
            }
            __syncthreads();
            if (tid == 0) {
                // lexicographic ratio test (fjsp_lp.cpp): a strictly sequential scan over the rows with a > eps
                int r = -1;
                const int n_el = s_nel;
                for (int q = 0; q < n_el; ++q) {
                    const int i = elig_rows[q];
                    const double a = cola[i];
                    if (r < 0) { r = i; continue; }
                    const double ar = cola[r];
                    const double vi = colv[i], vr = colv[r];
                    const double tol = 1e-12 * (fabs(vr) > 1.0 ? fabs(vr) : 1.0);
                    if (vi < vr - tol) { r = i; continue; }
                    if (vi > vr + tol) continue;
                    for (int c = nv; c < nv + nr; ++c) {
                        const double wi = at(i, c) / a, wr = at(r, c) / ar;
                        if (wi < wr) { r = i; break; }
                        if (wi > wr) break;
                    }
                }
                s_leave = r;
                if (r < 0) s_fail = 3;                                                     // "unbounded"
            }
            __syncthreads();
            const int r = s_leave;
            if (r < 0) break;
            // pivot: scale row r, eliminate column s from the other rows and from z
            const double piv = cola[r];
            double *rowr = &T[(size_t)r * nc];
            for (int j = tid; j < nc; j += kThreads) rowr[j] = rowr[j] / piv;
            __syncthreads();
            if (tid == 0) rowr[s] = 1.0;
            __syncthreads();
            const double fz = z[s];
            __syncthreads();
            {   // element (i, j): wave w takes the rows i = w, w + 4, ..., lane l the columns j = l, l + 64, ...
                const int w = tid >> 6, l = tid & 63;
                for (int i = w; i < nr; i += kThreads / 64) {
                    if (i == r) continue;
                    const double f = cola[i];
                    if (f == 0.0) continue;
                    double *rowi = &T[(size_t)i * nc];
                    for (int j = l; j < nc; j += 64) rowi[j] = rowi[j] - f * rowr[j];
                }
                if (fz != 0.0)
                    for (int j = tid; j < nc; j += kThreads) z[j] = z[j] - fz * rowr[j];
            }
            __syncthreads();
            for (int i = tid; i < nr; i += kThreads) {
                if (i == r || cola[i] == 0.0) continue;
                at(i, s) = 0.0;
                const double v = at(i, rhs);
                if (v < 0.0 && v > -1e-12) at(i, rhs) = 0.0;
            }
            if (tid == 0) { if (fz != 0.0) z[s] = 0.0; basis[r] = s; }
            __syncthreads();
        }
        __syncthreads();
        // ---- x out of the basis (values below 1e-11 are exact zeros, above 1 clamp to 1)
        for (int q = tid; q < KP * MP; q += kThreads) xout[q] = 0.0;
        __syncthreads();
        if (!s_fail) {
            // val[c] = rhs of the row whose basic variable is c: scatter through col_of's inverse -- every (k, m) looks its column up
            for (int q = tid; q < K * M; q += kThreads) {
                const int k = q / M, m = q % M;
                const uint16_t c = col_of[q];
                if (c == 0xFFFFu) continue;
                double v = 0.0;
                for (int i = 0; i < nr; ++i)
                    if (basis[i] == (int)c) v = at(i, rhs);
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
        if (s_fail && tid == 0) atomicOr(err, (uint32_t)s_fail);
        __syncthreads();
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
