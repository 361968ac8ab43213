// Device side of the C ABI (include/fjsp_amd.h): packs an instance set into the
// padded struct-of-arrays of fjsp_device.h, owns the HBM allocations of a batch
// of environments and launches the kernels of fjsp_kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fjsp_amd.h"
#include "fjsp_device.h"
#include "fjsp_host.h"
#include "fjsp_pyset.h"

using namespace fjsp;

struct fjsp_env {
    DevBatch b{};
    int device = 0;
    std::vector<void *> allocs;
    std::vector<int> inst_K, inst_M;   // per packed instance
    int64_t step_bytes = 0;
    // scratch for the non-fused rollout fallback
    uint8_t *d_done_scratch = nullptr;
};

namespace {

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        if (!hip_ok((expr), #expr)) return FJSP_E_HIP; \
    } while (0)

template <class T>
int upload(fjsp_env *e, const std::vector<T> &h, T **d) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, h.size() * sizeof(T) + 16));
    e->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = reinterpret_cast<T *>(p);
    return FJSP_OK;
}
template <class T>
int dalloc(fjsp_env *e, size_t n, T **d) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T) + 16));
    e->allocs.push_back(p);
    HIP_TRY(hipMemset(p, 0, n * sizeof(T)));
    *d = reinterpret_cast<T *>(p);
    return FJSP_OK;
}

// Python round(): half to even on the correctly rounded quotient (class_FJSSP.py:214-218)
long py_round(double v) { return (long)std::nearbyint(v); }

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int fjsp_env_create(const fjsp_instances *s, int32_t first, int32_t n_inst, int32_t n_envs, int32_t variant,
                    int32_t device, uint64_t rng_seed, fjsp_env **out) {
    if (!s || !out || first < 0 || n_inst <= 0 || n_envs <= 0 || (size_t)first + (size_t)n_inst > s->v.size()) {
        set_error("fjsp_env_create: bad arguments"); return FJSP_E_ARG;
    }
    if (variant != FJSP_VARIANT_SO_FJSSP && variant != FJSP_VARIANT_MO_FJSSP_DISCRETES) {
        set_error("fjsp_env_create: unknown variant"); return FJSP_E_ARG;
    }
    int Kmax = 0, Mmax = 0, Jmax = 0;
    for (int i = 0; i < n_inst; ++i) {
        const Instance &in = s->v[(size_t)first + i];
        if (!in.valid) { set_error("fjsp_env_create: instance not populated"); return FJSP_E_STATE; }
        if (!in.has_x) { set_error("fjsp_env_create: fluid solution missing (call fjsp_instances_solve_fluid)"); return FJSP_E_STATE; }
        if (in.S != 1) { set_error("multi-order instances (order arrival re-solves the LP mid-episode) are not supported by the kernels yet"); return FJSP_E_UNSUPPORTED; }
        if (in.K > kWave * kMaxKC) { set_error("more than 256 operation types"); return FJSP_E_UNSUPPORTED; }
        if (in.M > kMaxM) { set_error("more than 32 machines"); return FJSP_E_UNSUPPORTED; }
        const int nj = in.jobs_of_order(0);
        if (nj > 65535) { set_error("more than 65535 jobs"); return FJSP_E_UNSUPPORTED; }
        for (int r = 0; r < in.R; ++r)
            if (in.Jr[r] > 255) { set_error("more than 255 operations in a kind"); return FJSP_E_UNSUPPORTED; }
        for (int v : in.p)
            if (v > 65535) { set_error("processing time above 65535"); return FJSP_E_UNSUPPORTED; }
        Kmax = std::max(Kmax, in.K); Mmax = std::max(Mmax, in.M); Jmax = std::max(Jmax, nj);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device visible: the environment kernels need an MI355X (there is no CPU path)");
        return FJSP_E_HIP;
    }
    if (device < 0 || device >= ndev) { set_error("fjsp_env_create: device index out of range"); return FJSP_E_ARG; }
    DeviceGuard guard(device);

    auto *e = new fjsp_env();
    e->device = device;
    DevBatch &b = e->b;
    b.N = n_envs; b.n_inst = n_inst;
    b.KC = Kmax <= 64 ? 1 : (Kmax <= 128 ? 2 : 4);
    b.KP = b.KC * kWave;
    b.MP = Mmax;
    b.JP = ((Jmax + 63) / 64) * 64;
    b.variant = variant;
    b.n_obs = variant == FJSP_VARIANT_SO_FJSSP ? 10 : 9;
    b.n_static = variant == FJSP_VARIANT_SO_FJSSP ? 0 : 7;
    b.state_size = b.n_static + 2 * b.n_obs;
    b.rng_seed = rng_seed;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP, JP = (size_t)b.JP, NI = (size_t)n_inst;

    std::vector<InstHeader> ihdr(NI);
    std::vector<uint32_t> kA(NI * KP, 0), kB(NI * KP, 0), elig(NI * KP, 0), first4(NI * KP, 0), jinfo(NI * JP, 0);
    std::vector<uint16_t> p(NI * MP * KP, 0);
    std::vector<double> x(NI * MP * KP, 0.0), sstate(NI * 8, 0.0);
    std::vector<int32_t> due(NI * JP, 0);
    double bytes_acc = 0.0;
    for (size_t i = 0; i < NI; ++i) {
        const Instance &in = s->v[(size_t)first + i];
        const int nj = in.jobs_of_order(0);
        ihdr[i] = InstHeader{in.K, in.M, in.R, nj};
        e->inst_K.push_back(in.K); e->inst_M.push_back(in.M);
        int jbeg = 0;
        for (int r = 0; r < in.R; ++r) {
            const int cnt = in.count[r];
            // class_FJSSP.py:214-218: r_due = round(delivery * J_r / N_r); due(n) = round(r_due * n / N_r)
            const long r_due = py_round((double)((long)in.delivery[0] * in.Jr[r]) / (double)cnt);
            for (int n = 0; n < cnt; ++n) {
                due[i * JP + jbeg + n] = (int32_t)py_round((double)(r_due * n) / (double)cnt);
                jinfo[i * JP + jbeg + n] = (uint32_t)in.koff[r] | ((uint32_t)in.Jr[r] << 16);
            }
            for (int j = 0; j < in.Jr[r]; ++j) {
                const int k = in.koff[r] + j;
                kA[i * KP + k] = (uint32_t)jbeg | ((uint32_t)cnt << 16);
                kB[i * KP + k] = (uint32_t)j | ((uint32_t)in.Jr[r] << 8) | ((uint32_t)(r & 0xFF) << 16) |
                                 ((uint32_t)((j == in.Jr[r] - 1 ? 1u : 0u) | 2u) << 24);
                uint32_t em = 0;
                for (int m = 0; m < in.M; ++m) {
                    const int pv = in.p[(size_t)k * in.M + m];
                    if (pv > 0) em |= 1u << m;
                    p[(i * MP + m) * KP + k] = (uint16_t)pv;
                    x[(i * MP + m) * KP + k] = in.x[(size_t)k * in.M + m];
                }
                elig[i * KP + k] = em;
                uint32_t f4 = 0;
                for (int q = 0; q < in.elig_n[k] && q < 4; ++q) f4 |= (uint32_t)in.elig_list[(size_t)k * in.M + q] << (8 * q);
                first4[i * KP + k] = f4;
            }
            jbeg += cnt;
        }
        if (variant == FJSP_VARIANT_MO_FJSSP_DISCRETES) {
            // MO_FJSSP_discretes.py:55-64 static_state_extract (math.pow(v, 2) is libm pow on the host)
            long ns = 0, js = 0;
            for (int r = 0; r < in.R; ++r) { ns += in.count[r]; js += in.Jr[r]; }
            const double N_ave = (double)ns / (double)in.R, J_ave = (double)js / (double)in.R;
            double a = 0.0, c2 = 0.0;
            for (int r = 0; r < in.R; ++r) a = a + std::pow((double)in.count[r] - N_ave, 2.0);
            for (int r = 0; r < in.R; ++r) c2 = c2 + std::pow((double)in.Jr[r] - J_ave, 2.0);
            double *ss = &sstate[i * 8];
            ss[0] = in.ddt; ss[1] = (double)in.M; ss[2] = (double)in.R; ss[3] = N_ave;
            ss[4] = std::sqrt(a / (double)in.R); ss[5] = J_ave; ss[6] = std::sqrt(c2 / (double)in.R);
        }
        // algorithmic HBM bytes of one env-step (DESIGN.md "bytes per env-step"):
        //   static per-k rows (kinfoA/B, elig, fmask u32; rate_sum, time_sum f64)      K * 32
        //   job table read (due, jinfo, jst) + jst write-back                           njobs * 16
        //   machine lanes tend/mjob read + write                                        M * 16
        //   EnvScalars read + write                                                     2 * 144
        //   column gather at k_sel (p u16, un/arr/rate f64) + un read-modify-write      M * 26 + 8
        //   actions in, state/reward/done out                                           2 + S*8 + 8 + 1
        bytes_acc += in.K * 32.0 + nj * 16.0 + in.M * 16.0 + 288.0 + in.M * 26.0 + 8.0 + 2.0 + b.state_size * 8.0 + 9.0;
    }
    e->step_bytes = (int64_t)(bytes_acc / (double)NI + 0.5);

    int rc = FJSP_OK;
    InstHeader *d_ihdr = nullptr; uint32_t *d_kA = nullptr, *d_kB = nullptr, *d_elig = nullptr, *d_jinfo = nullptr, *d_f4 = nullptr;
    uint16_t *d_p = nullptr; double *d_x = nullptr, *d_ss = nullptr; int32_t *d_due = nullptr;
#define TRY(call) do { rc = (call); if (rc != FJSP_OK) { fjsp_env_destroy(e); return rc; } } while (0)
    TRY(upload(e, ihdr, &d_ihdr)); TRY(upload(e, kA, &d_kA)); TRY(upload(e, kB, &d_kB)); TRY(upload(e, elig, &d_elig));
    TRY(upload(e, jinfo, &d_jinfo)); TRY(upload(e, p, &d_p)); TRY(upload(e, x, &d_x)); TRY(upload(e, sstate, &d_ss));
    TRY(upload(e, due, &d_due)); TRY(upload(e, first4, &d_f4));
    b.ihdr = d_ihdr; b.kinfoA = d_kA; b.kinfoB = d_kB; b.elig = d_elig; b.jinfo = d_jinfo; b.p = d_p; b.x = d_x;
    b.sstate = d_ss; b.due = d_due; b.efirst4 = d_f4;
    TRY(dalloc(e, NI * KP, &b.fmask));
    TRY(dalloc(e, NI * MP * KP, &b.rate)); TRY(dalloc(e, NI * MP * KP, &b.arr));
    TRY(dalloc(e, NI * KP, &b.rate_sum)); TRY(dalloc(e, NI * KP, &b.time_sum));
    const size_t N = (size_t)n_envs;
    TRY(dalloc(e, N, &b.scal)); TRY(dalloc(e, N * MP, &b.tend)); TRY(dalloc(e, N * MP, &b.mjob));
    TRY(dalloc(e, N * JP, &b.jst)); TRY(dalloc(e, N * MP * KP, &b.un));
    TRY(dalloc(e, N, &e->d_done_scratch));
#undef TRY
    if (launch_fluid_tables(b, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) {
        set_error("fluid_tables_kernel launch failed");
        fjsp_env_destroy(e);
        return FJSP_E_HIP;
    }
    // every env starts done so that step() before reset() is flagged, like the
    // reference's uninitialised object would fail
    {
        std::vector<EnvScalars> init(N);
        std::memset(init.data(), 0, N * sizeof(EnvScalars));
        for (auto &sc : init) sc.done = 1;
        if (hipMemcpy(b.scal, init.data(), N * sizeof(EnvScalars), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("scalars upload failed"); fjsp_env_destroy(e); return FJSP_E_HIP;
        }
    }
    *out = e;
    return FJSP_OK;
}

void fjsp_env_destroy(fjsp_env *e) {
    if (!e) return;
    {
        DeviceGuard guard(e->device);
        for (void *p : e->allocs) (void)hipFree(p);
    }
    delete e;
}

int fjsp_env_num_envs(const fjsp_env *e) { return e ? e->b.N : 0; }
int fjsp_env_state_size(const fjsp_env *e) { return e ? e->b.state_size : 0; }
int fjsp_env_device(const fjsp_env *e) { return e ? e->device : -1; }
int64_t fjsp_env_step_bytes(const fjsp_env *e) { return e ? e->step_bytes : 0; }

int fjsp_env_reset(fjsp_env *e, const uint8_t *d_mask, double *d_state, void *stream) {
    if (!e) { set_error("fjsp_env_reset: null env"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    if (launch_reset(e->b, d_mask, d_state, (hipStream_t)stream) != 0) { set_error("reset_kernel launch failed"); return FJSP_E_HIP; }
    return FJSP_OK;
}

int fjsp_env_step(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset, double *d_state,
                  double *d_reward, uint8_t *d_done, void *stream) {
    if (!e || !d_actions) { set_error("fjsp_env_step: null argument"); return FJSP_E_ARG; }
    if (e->b.variant != FJSP_VARIANT_SO_FJSSP) { set_error("fjsp_env_step: MO variant kernels not built yet"); return FJSP_E_UNSUPPORTED; }
    (void)d_mo;
    DeviceGuard guard(e->device);
    if (launch_step(e->b, d_actions, autoreset, d_state, d_reward, d_done, nullptr, (hipStream_t)stream) != 0) {
        set_error("step_kernel launch failed"); return FJSP_E_HIP;
    }
    return FJSP_OK;
}

int fjsp_env_rollout(fjsp_env *e, const uint8_t *d_actions, int32_t T, int16_t *d_trace_km, double *d_reward,
                     double *d_state_last, void *stream) {
    if (!e || !d_actions || T <= 0) { set_error("fjsp_env_rollout: bad arguments"); return FJSP_E_ARG; }
    if (e->b.variant != FJSP_VARIANT_SO_FJSSP) { set_error("fjsp_env_rollout: MO variant kernels not built yet"); return FJSP_E_UNSUPPORTED; }
    DeviceGuard guard(e->device);
    hipStream_t st = (hipStream_t)stream;
    if (rollout_lds_bytes(e->b) <= 64 * 1024) {
        if (launch_rollout(e->b, d_actions, T, d_trace_km, d_reward, d_state_last, st) != 0) {
            set_error("rollout_kernel launch failed"); return FJSP_E_HIP;
        }
        return FJSP_OK;
    }
    // instance too large for the LDS-resident fused kernel: T step launches on the same stream
    const size_t N = (size_t)e->b.N;
    for (int s2 = 0; s2 < T; ++s2) {
        if (launch_step(e->b, d_actions + (size_t)s2 * N * 2, 0, d_state_last, d_reward ? d_reward + (size_t)s2 * N : nullptr,
                        e->d_done_scratch, d_trace_km ? d_trace_km + (size_t)s2 * N * 2 : nullptr, st) != 0) {
            set_error("step_kernel launch failed"); return FJSP_E_HIP;
        }
    }
    return FJSP_OK;
}

int fjsp_env_read(fjsp_env *e, int64_t *d_delay_time_sum, int32_t *d_makespan, int32_t *d_completion,
                  int32_t *d_step_time, int32_t *d_step_count, uint8_t *d_done, uint32_t *d_status, void *stream) {
    if (!e) { set_error("fjsp_env_read: null env"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    if (launch_read(e->b, d_delay_time_sum, d_makespan, d_completion, d_step_time, d_step_count, d_done, d_status,
                    (hipStream_t)stream) != 0) { set_error("read_kernel launch failed"); return FJSP_E_HIP; }
    return FJSP_OK;
}

int fjsp_env_machine_time_end(fjsp_env *e, int32_t *d_tend, int32_t m_stride, void *stream) {
    if (!e || !d_tend || m_stride < e->b.MP) { set_error("fjsp_env_machine_time_end: bad arguments"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemcpy2DAsync(d_tend, (size_t)m_stride * 4, e->b.tend, (size_t)e->b.MP * 4, (size_t)e->b.MP * 4,
                             (size_t)e->b.N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FJSP_OK;
}

int fjsp_pyset_and_order(uint32_t idle_mask, const int32_t *machines, int32_t n, int32_t ascending, int32_t *out) {
    if (!out || (n > 0 && !machines) || n < 0 || n > 32) { set_error("fjsp_pyset_and_order: bad arguments"); return FJSP_E_ARG; }
    uint32_t bm = 0, f4 = 0;
    for (int q = 0; q < n; ++q) {
        if (machines[q] < 0 || machines[q] > 31) { set_error("fjsp_pyset_and_order: machine index out of range"); return FJSP_E_ARG; }
        bm |= 1u << machines[q];
        if (q < 4) f4 |= (uint32_t)machines[q] << (8 * q);
    }
    const CandList c = pyset_and(idle_mask, bm, f4, ascending != 0);
    for (int i = 0; i < c.n; ++i) out[i] = cand_at(c, i);
    return c.n;
}

int fjsp_env_fluid_tables(fjsp_env *e, int32_t i, double *h_rate, double *h_arr, double *h_rate_sum, double *h_time_sum) {
    if (!e || i < 0 || i >= e->b.N) { set_error("fjsp_env_fluid_tables: bad arguments"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    const int inst = i % e->b.n_inst;
    const int K = e->inst_K[(size_t)inst], M = e->inst_M[(size_t)inst];
    const size_t KP = (size_t)e->b.KP, MP = (size_t)e->b.MP;
    std::vector<double> rate(MP * KP), arr(MP * KP), rs(KP), ts(KP);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rate.data(), e->b.rate + (size_t)inst * MP * KP, MP * KP * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(arr.data(), e->b.arr + (size_t)inst * MP * KP, MP * KP * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rs.data(), e->b.rate_sum + (size_t)inst * KP, KP * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts.data(), e->b.time_sum + (size_t)inst * KP, KP * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) {
        for (int m = 0; m < M; ++m) {
            if (h_rate) h_rate[(size_t)k * M + m] = rate[(size_t)m * KP + k];
            if (h_arr) h_arr[(size_t)k * M + m] = arr[(size_t)m * KP + k];
        }
        if (h_rate_sum) h_rate_sum[k] = rs[(size_t)k];
        if (h_time_sum) h_time_sum[k] = ts[(size_t)k];
    }
    return FJSP_OK;
}

}  // extern "C"
