// Device side of the C ABI (include/fjsp_amd.h): packs an instance set into the
// padded struct-of-arrays of fjsp_device.h, owns the HBM allocations of a batch
// of environments and launches the kernels of fjsp_kernels.hip.
#include <hip/hip_runtime.h>

#include <chrono>
#include <deque>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/fjsp_amd.h"
#include "fjsp_device.h"
#include "fjsp_host.h"
#include "fjsp_pyset.h"
#include "fjsp_policy.h"

using namespace fjsp;

// Order-arrival LPs repeat: environments that play the same instance and reach an arrival in the same situation (the
// same unprocessed / waiting counts per operation type -- always the case when the shop had run empty and the clock
// jumped to the arrival, SO_FJSSP.py:228-231) pose the same LP.  The LP is a pure function of (instance, Q, n_now), so
// its solution is remembered (same bits as a fresh solve).
struct LpCache {
    std::mutex mu;
    std::unordered_map<std::string, std::vector<double>> map;
    int64_t hits = 0, misses = 0;
    static std::string key(int inst, const uint16_t *lpq, size_t KP, int K) {
        std::string k(sizeof(int) + (size_t)K * 4, '\0');
        std::memcpy(&k[0], &inst, sizeof(int));
        std::memcpy(&k[sizeof(int)], lpq, (size_t)K * 2);
        std::memcpy(&k[sizeof(int) + (size_t)K * 2], lpq + KP, (size_t)K * 2);
        return k;
    }
    bool find(const std::string &k, std::vector<double> &x) {
        std::lock_guard<std::mutex> g(mu);
        auto it = map.find(k);
        if (it == map.end()) { ++misses; return false; }
        ++hits; x = it->second;
        return true;
    }
    // bounded by bytes (keys + solutions; 256 MiB): a full memo stops taking entries -- the LPs it misses are solved
    size_t bytes = 0;
    static constexpr size_t kMaxBytes = (size_t)256 << 20;
    void put(const std::string &k, const std::vector<double> &x) {
        std::lock_guard<std::mutex> g(mu);
        const size_t add = k.size() + x.size() * sizeof(double) + 64;
        if (bytes + add > kMaxBytes) return;
        if (map.emplace(k, x).second) bytes += add;
    }
};

struct fjsp_env {
    DevBatch b{};
    int device = 0;
    std::vector<void *> allocs;
    std::vector<int> inst_K, inst_M;   // per packed instance
    int64_t step_bytes = 0;
    // scratch for the non-fused rollout fallback
    uint8_t *d_done_scratch = nullptr;
    // multi-order service (host LP at order arrivals)
    const fjsp_instances *src = nullptr;
    int first = 0;
    // pinned host staging of the service: [0] = count, then env ids | LP inputs per slot | solutions per slot
    uint32_t *h_pending = nullptr;
    uint16_t *h_lp_in = nullptr;
    double *h_lp_x = nullptr;
    // asynchronous arrival service (fjsp_env_step_async): a ring of batches of parked envs on their way through
    // D2H copy -> host LPs (dispatcher thread) -> upload -> arrival_kernel, while the other envs keep stepping
    struct AsyncBatch *ring = nullptr;
    int ring_n = 0, ring_next = 0;
    struct LpWorkers *workers = nullptr;
    hipStream_t copy_stream = nullptr;  // the parked envs' ids / LP inputs leave on their own stream: the next step launch does not wait for them
    hipEvent_t ev_step = nullptr;
    LpCache lp_cache;
    uint32_t *d_resume_ids = nullptr;   // [N] device list handed to arrival_kernel
    double *d_resume_x = nullptr;       // [N][KP][MP]
    int64_t async_parked = 0;           // envs currently parked (host view)
    bool failed = false;        // the arrival service failed mid-step: parked envs are in limbo, the handle refuses further steps
    int lp_threads = 0;         // 0 = default (min(host cores, 16))
    int64_t lp_solves = 0;      // order-arrival LPs solved so far (host service)
    // device LP service (fjsp_lp_device.hip): chosen at create time when the largest tableau of the batch fits the CU's LDS
    bool lp_device = false;
    size_t lp_lds = 0;
    uint32_t *d_lp_err = nullptr;                // [0] nonzero: an LP failed on the device (reported at the next synchronising call)
    unsigned long long *d_lp_solved = nullptr;   // LPs solved on the device so far
    struct LpPool *pool = nullptr;
};

// Persistent worker threads of the order-arrival LP service: run(n, fn) calls fn(q) for q in [0, n) on the
// workers and the caller, returning when all are done.
struct LpPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(uint32_t)> fn;
    std::atomic<uint32_t> next{0};
    uint32_t n = 0, generation = 0;
    int active = 0;
    bool stop = false;

    explicit LpPool(int n_workers) {
        for (int t = 0; t < n_workers; ++t) workers.emplace_back([this] { loop(); });
    }
    ~LpPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
    void drain() {
        for (;;) {
            const uint32_t q = next.fetch_add(1);
            if (q >= n) return;
            fn(q);
        }
    }
    void loop() {
        uint32_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            drain();
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }
    void run(uint32_t count, std::function<void(uint32_t)> f) {
        {
            std::lock_guard<std::mutex> g(mu);
            fn = std::move(f); n = count; next.store(0); active = (int)workers.size(); ++generation;
        }
        cv_work.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

// One launch's parked environments on their way through the asynchronous arrival service.
struct AsyncBatch {
    enum State { FREE, HEAD_COPY, TAIL_COPY, SOLVING, SOLVED, UPLOADING };
    uint32_t *d_count = nullptr;      // device staging the parking waves write: [0] = count, [1 + slot] = env id
    uint16_t *d_lp_in = nullptr;      //                                         [slot][2][KP] LP inputs (Q, n_now)
    uint32_t *h_ids = nullptr;        // pinned mirrors
    uint16_t *h_lp_in = nullptr;
    double *h_x = nullptr;            // pinned [slot][KP][MP] solutions
    hipEvent_t ev_head = nullptr, ev_tail = nullptr, ev_up = nullptr;
    State state = FREE;
    uint32_t n = 0, cap = 0;          // parked envs of this batch; capacity of the pinned mirrors (grown on demand)
    std::atomic<int> solved{0};       // 1 = every LP solved, -1 = a solve failed
    std::atomic<int> left{0}, bad{0};
    std::mutex err_mu;
    std::string err;
};

// Worker threads of the asynchronous service: ONE queue of single LPs across all batches in flight, so that the
// threads stay busy whatever the batch sizes are; a batch is solved when its last LP is.
struct LpWorkers {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<AsyncBatch *, uint32_t>> tasks;
    std::function<bool(AsyncBatch *, uint32_t)> solve;      // false: the LP failed (message left in the batch)
    bool stop = false;
    LpWorkers(int n, std::function<bool(AsyncBatch *, uint32_t)> f) : solve(std::move(f)) {
        for (int t = 0; t < n; ++t) th.emplace_back([this] { loop(); });
    }
    ~LpWorkers() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void submit(AsyncBatch *a) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (uint32_t q = 0; q < a->n; ++q) tasks.emplace_back(a, q);
        }
        cv.notify_all();
    }
    void loop() {
        for (;;) {
            std::pair<AsyncBatch *, uint32_t> t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !tasks.empty(); });
                if (tasks.empty()) return;              // (stop: drain first)
                t = tasks.front();
                tasks.pop_front();
            }
            if (!solve(t.first, t.second)) t.first->bad.store(1);
            if (t.first->left.fetch_sub(1) == 1) t.first->solved.store(t.first->bad.load() ? -1 : 1);
        }
    }
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
        // (the common case -- the caller is already on the batch's device -- costs one hipGetDevice: the per-step
        // calls are launch-bound on the host)
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int fjsp_env_create(const fjsp_instances *s, int32_t first, int32_t n_inst, int32_t n_envs, int32_t variant,
                    int32_t device, uint64_t rng_seed, fjsp_env **out) {
    // SO_DFJSP.py = SO_FJSSP.py over class_FJSP.py: the same kernels, the order's delivery time as every job's due date
    const bool class_fjsp = variant == FJSP_VARIANT_SO_DFJSP;
    if (class_fjsp) variant = FJSP_VARIANT_SO_FJSSP;
    if (!s || !out || first < 0 || n_inst <= 0 || n_envs <= 0 || (size_t)first + (size_t)n_inst > s->v.size()) {
        set_error("fjsp_env_create: bad arguments"); return FJSP_E_ARG;
    }
    const bool dyn = variant == FJSP_VARIANT_MO_DFJSP;
    if (variant != FJSP_VARIANT_SO_FJSSP && variant != FJSP_VARIANT_SO_SFJSP && variant != FJSP_VARIANT_MO_FJSSP_DISCRETES && !dyn) {
        set_error("fjsp_env_create: unknown variant"); return FJSP_E_ARG;
    }
    int Kmax = 0, Mmax = 0, Jmax = 0, Smax = 1, Rmax = 0, Bmax = 1;
    bool single_job = true;
    for (int i = 0; i < n_inst; ++i) {
        const Instance &in = s->v[(size_t)first + i];
        if (!in.valid) { set_error("fjsp_env_create: instance not populated"); return FJSP_E_STATE; }
        if (!in.has_x) { set_error("fjsp_env_create: fluid solution missing (call fjsp_instances_solve_fluid)"); return FJSP_E_STATE; }
        if (class_fjsp) {
            for (int m = 0; m < in.M; ++m) {
                bool any = false;
                for (int k = 0; k < in.K; ++k) any = any || in.p[(size_t)k * in.M + m] > 0;
                // class_FJSP.py:159 divides by len(kind_task_tuple); with at least one operation type n + 1e-18 == n in f64
                if (!any) { set_error("SO_DFJSP: a machine with no eligible operation (ZeroDivisionError in the reference)"); return FJSP_E_UNSUPPORTED; }
            }
        }
        if (dyn) {
            if (!in.has_dynamic) { set_error("MO_DFJSP needs machine data (machine_data.csv / fjsp_instances_set_dynamic)"); return FJSP_E_STATE; }
            int nb = 0;
            for (int m = 0; m < in.M; ++m) {
                nb += in.bk_n[m];
                bool any = false;
                for (int k = 0; k < in.K; ++k) any = any || in.p[(size_t)k * in.M + m] > 0;
                // Machine.gap_ave divides by len(kind_task_tuple) with no epsilon (class_MODFJSP.py:158-159)
                if (!any) { set_error("MO_DFJSP: a machine with no eligible operation (ZeroDivisionError in the reference)"); return FJSP_E_UNSUPPORTED; }
            }
            if (nb > 65535) { set_error("more than 65535 breakdown windows"); return FJSP_E_UNSUPPORTED; }
            Bmax = std::max(Bmax, nb);
            for (size_t q = 0; q < in.p.size(); ++q) {
                if (in.power[q] < 0 || in.power[q] > 65535) { set_error("power above 65535"); return FJSP_E_UNSUPPORTED; }
                if ((long long)in.power[q] * in.p[q] > 0x7fffffffLL) { set_error("energy of one operation above 2^31"); return FJSP_E_UNSUPPORTED; }
            }
        }
        if (in.S != 1 && variant != FJSP_VARIANT_SO_FJSSP && !dyn) { set_error("only SO_FJSSP handles order arrivals (the subclasses are single-order, SO_SFJSP.py:20 / MO_FJSSP_discretes.py:21)"); return FJSP_E_UNSUPPORTED; }
        if (in.S > 64) { set_error("more than 64 orders"); return FJSP_E_UNSUPPORTED; }
        Smax = std::max(Smax, in.S); Rmax = std::max(Rmax, in.R);
        if (in.K > kWave * kMaxKC) { set_error("more than 256 operation types"); return FJSP_E_UNSUPPORTED; }
        if (in.M > kMaxM) { set_error("more than 32 machines"); return FJSP_E_UNSUPPORTED; }
        const int nj = in.jobs_total();
        if (nj > 65535) { set_error("more than 65535 jobs"); return FJSP_E_UNSUPPORTED; }
        {
            long ntasks = 0;
            for (int so = 0; so < in.S; ++so)
                for (int r = 0; r < in.R; ++r) ntasks += (long)in.count[(size_t)so * in.R + r] * in.Jr[r];
            if (ntasks > 65535) { set_error("more than 65535 operations"); return FJSP_E_UNSUPPORTED; }
        }
        for (int r = 0; r < in.R; ++r)
            if (in.Jr[r] > 255) { set_error("more than 255 operations in a kind"); return FJSP_E_UNSUPPORTED; }
        long long pmax = 0;
        for (int v : in.p) {
            if (v > 65535) { set_error("processing time above 65535"); return FJSP_E_UNSUPPORTED; }
            pmax = std::max<long long>(pmax, v);
        }
        {
            // The kernels keep the clock, the machines' time_end and the per-kind tardiness sums in 32-bit integers
            // (Python integers do not wrap).  Worst case of the clock: every operation in sequence at the largest
            // processing time, after the last order arrival, stretched by every breakdown window; worst case of a
            // per-kind sum: every job of the kind late by that much.
            long long ntasks = 0, jobs_kind_max = 0, t_arr_max = 0, bk_total = 0, due_max = 0;
            for (int r = 0; r < in.R; ++r) {
                long long jk = 0;
                for (int so = 0; so < in.S; ++so) { jk += in.count[(size_t)so * in.R + r]; ntasks += (long long)in.count[(size_t)so * in.R + r] * in.Jr[r]; }
                jobs_kind_max = std::max(jobs_kind_max, jk);
            }
            for (int so = 0; so < in.S; ++so) {
                t_arr_max = std::max<long long>(t_arr_max, in.arrive[so]);
                due_max = std::max<long long>(due_max, std::llabs((long long)in.delivery[so]));
            }
            if (dyn)
                for (size_t q = 0; q + 1 < in.bk.size(); q += 2) bk_total += std::max(0, in.bk[q + 1] - in.bk[q]);
            const long long clock_max = t_arr_max + ntasks * pmax + bk_total;
            if (clock_max + due_max > 0x7fffffffLL || jobs_kind_max * (clock_max + due_max) > 0x7fffffffLL) {
                set_error("instance too long for the kernels' 32-bit clocks: (last arrival + operations x max processing time + "
                          "breakdown windows) x jobs per kind must stay below 2^31");
                return FJSP_E_UNSUPPORTED;
            }
        }
        for (int r = 0; r < in.R; ++r) single_job = single_job && in.S == 1 && in.R <= 255 && in.count[(size_t)r] == 1;
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
    b.variant = variant;
    b.n_obs = variant == FJSP_VARIANT_SO_FJSSP ? 10 : (dyn ? 15 : 9);
    b.n_static = variant == FJSP_VARIANT_MO_FJSSP_DISCRETES ? 7 : 0;
    b.state_size = b.n_static + 2 * b.n_obs;
    b.rng_seed = rng_seed;
    b.mord = (Smax > 1 || dyn) ? 1 : 0; b.SP = Smax; b.RP = Rmax;     // MO_DFJSP always runs on the per-env fluid tables
    b.single_job = (single_job && !b.mord) ? 1 : 0;
    b.kmax = Kmax;
    // one 16-lane row per environment (fjsp_group.hip): FJSP_STEP_IMPL=wave keeps such batches on the one-wave-per-environment kernels
    {
        const char *impl = getenv("FJSP_STEP_IMPL");
        b.grp = (b.single_job && Kmax <= 64 && Mmax <= 8 && Jmax <= 15 &&
                 (variant == FJSP_VARIANT_SO_FJSSP || variant == FJSP_VARIANT_MO_FJSSP_DISCRETES) && !(impl && strcmp(impl, "wave") == 0)) ? 1 : 0;
    }
    // job words per record: a multiple of 64, or 16 in row-kernel batches (at most 15 jobs) -- the dynamic record of such an
    // environment then spans three 128-byte lines instead of five
    b.JP = b.grp ? 16 : ((Jmax + 63) / 64) * 64;
    b.jcap = std::min(b.JP, (Jmax + 15) / 16 * 16);
    if (step_lds_bytes(b) > 160 * 1024) {
        // four environments per workgroup keep their job tables (8 bytes per job) in the CU's 160 KB of LDS
        char msg[200];
        snprintf(msg, sizeof(msg), "instance too large for the kernels' LDS staging: %d jobs need %zu bytes per workgroup, the limit is "
                 "163840 (about %d jobs at this shape)", Jmax, step_lds_bytes(b), (int)((160 * 1024 / 4 - (30 + 3 * b.KP) * 8 - 256) / 8 / 64 * 64));
        set_error(msg);
        delete e;
        return FJSP_E_UNSUPPORTED;
    }
    e->src = s; e->first = first;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP, JP = (size_t)b.JP, NI = (size_t)n_inst, N = (size_t)n_envs;
    // ---- record layouts (fjsp_device.h)
    Layout &L = b.L;
    {
        size_t o = 64;                                   // InstHeader, padded
        auto take = [&](size_t bytes, size_t align) { o = (o + align - 1) / align * align; size_t at = o; o += bytes; return (uint32_t)at; };
        L.i_kA = take(KP * 4, 4); L.i_kB = take(KP * 4, 4); L.i_elig = take(KP * 4, 4); L.i_fmask = take(KP * 4, 4);
        L.i_f4 = take(KP * 4, 4); L.i_rsum = take(KP * 8, 8); L.i_tsum = take(KP * 8, 8);
        L.i_due = take(JP * 4, 4); L.i_jinfo = take(JP * 4, 4); L.i_p = take(MP * KP * 2, 4);
        L.i_x = take(MP * KP * 8, 8); L.i_col = take(MP * KP * 16, 16);
        L.i_ss = take(64, 8); L.i_obs0 = take(128, 8);
        L.i_oarr = take((size_t)Smax * 4, 4); L.i_ocnt = take((size_t)Smax * Rmax * 2, 4);
        if (dyn) {
            L.i_pw = take(MP * KP * 2, 4); L.i_ipw = take(MP * 4, 4); L.i_bkoff = take((MP + 1) * 2, 4);
            L.i_bk = take((size_t)Bmax * 8, 8);
        }
        L.i_op = b.grp ? take(2048 + 128, 256) : 0u;
        L.i_colm = b.grp ? take(MP * 64 * 16, 128) : 0u;
        L.i_op8 = b.grp ? take(64 * 8, 128) : 0u;
        L.i_stride = (uint32_t)((o + 255) / 256 * 256);
        o = 192;                                         // EnvScalars (144 B), padded
        L.e_tend = take(MP * 4, 4); L.e_mjob = take(MP * 4, 4); L.e_jst = take(JP * 4, 4); L.e_un = take(b.single_job ? 8 : MP * KP * 8, 8); L.e_asg = take(KP, 4);
        if (b.mord) {
            L.e_q0 = take(KP * 4, 4); L.e_fmask = take(KP * 4, 4); L.e_rsum = take(KP * 8, 8); L.e_tsum = take(KP * 8, 8);
            L.e_col = take(MP * KP * 16, 16); L.e_lpq = take(KP * 4 + 8, 4);
        }
        if (dyn) L.e_dyn = take(sizeof(DynScalars) + MP * 4, 8);
        L.e_stats = b.single_job ? 0u : take(KP * 64, 64);
        L.e_stride = (uint32_t)((o + 127) / 128 * 128);
    }
    {   // the kernels compute these offsets themselves (FixedOffsets, fjsp_device.h): the two must agree
        using F = FixedOffsets;
        const uint32_t kp = (uint32_t)KP, mp = (uint32_t)MP, jp = (uint32_t)JP;
        const bool same = L.i_kA == F::i_kA(kp) && L.i_kB == F::i_kB(kp) && L.i_elig == F::i_elig(kp) && L.i_fmask == F::i_fmask(kp) &&
                          L.i_f4 == F::i_f4(kp) && L.i_rsum == F::i_rsum(kp) && L.i_tsum == F::i_tsum(kp) && L.i_due == F::i_due(kp) &&
                          L.e_tend == F::e_tend() && L.e_mjob == F::e_mjob(mp) && L.e_jst == F::e_jst(mp) && L.e_un == F::e_un(mp, jp) &&
                          L.e_asg == F::e_asg(mp, jp, kp, b.single_job != 0) &&
                          (b.mord || dyn || L.e_stride == F::e_stride_plain(mp, jp, kp, b.single_job != 0));
        if (!same) { delete e; set_error("fjsp_env_create: record layout and FixedOffsets disagree (internal error)"); return FJSP_E_UNSUPPORTED; }
    }
    std::vector<unsigned char> islab(NI * L.i_stride, 0);
    auto ip = [&](size_t i, uint32_t off) { return islab.data() + i * L.i_stride + off; };
    double bytes_acc = 0.0;
    for (size_t i = 0; i < NI; ++i) {
        const Instance &in = s->v[(size_t)first + i];
        const int nj = in.jobs_total();
        *reinterpret_cast<InstHeader *>(ip(i, 0)) = InstHeader{in.K, in.M, in.R | (b.mord ? in.S << 16 : 0), nj};
        {
            int32_t *oarr = reinterpret_cast<int32_t *>(ip(i, L.i_oarr));
            uint16_t *ocnt = reinterpret_cast<uint16_t *>(ip(i, L.i_ocnt));
            for (int so = 0; so < in.S; ++so) {
                oarr[so] = in.arrive[so];
                for (int r = 0; r < in.R; ++r) ocnt[(size_t)so * Rmax + r] = (uint16_t)in.count[(size_t)so * in.R + r];
            }
        }
        e->inst_K.push_back(in.K); e->inst_M.push_back(in.M);
        uint32_t *kA = reinterpret_cast<uint32_t *>(ip(i, L.i_kA)), *kB = reinterpret_cast<uint32_t *>(ip(i, L.i_kB));
        uint32_t *elig = reinterpret_cast<uint32_t *>(ip(i, L.i_elig)), *first4 = reinterpret_cast<uint32_t *>(ip(i, L.i_f4));
        uint32_t *jinfo = reinterpret_cast<uint32_t *>(ip(i, L.i_jinfo));
        int32_t *due = reinterpret_cast<int32_t *>(ip(i, L.i_due));
        uint16_t *p = reinterpret_cast<uint16_t *>(ip(i, L.i_p));
        double *x = reinterpret_cast<double *>(ip(i, L.i_x));
        int jbeg = 0;
        for (int r = 0; r < in.R; ++r) {
            // jobs of kind r over ALL orders, numbered in arrival order (Kind.number_start, class_FJSSP.py:212-217);
            // class_FJSSP.py:214-218: r_due = round(delivery_s * J_r / N_sr); due(n) = round(r_due * n / N_sr) with
            // the ABSOLUTE job number n
            int cnt = 0;
            for (int so = 0; so < in.S; ++so) {
                const int c_s = in.count[(size_t)so * in.R + r];
                const long r_due = py_round((double)((long)in.delivery[so] * in.Jr[r]) / (double)c_s);
                for (int n = cnt; n < cnt + c_s; ++n) {
                    due[jbeg + n] = (dyn || class_fjsp) ? in.delivery[so]                 // class_MODFJSP.py:224, class_FJSP.py:229
                                        : (int32_t)py_round((double)(r_due * n) / (double)c_s);
                    jinfo[jbeg + n] = (uint32_t)in.koff[r] | ((uint32_t)in.Jr[r] << 16);
                }
                cnt += c_s;
            }
            for (int j = 0; j < in.Jr[r]; ++j) {
                const int k = in.koff[r] + j;
                kA[k] = (uint32_t)jbeg | ((uint32_t)cnt << 16);
                kB[k] = (uint32_t)j | ((uint32_t)in.Jr[r] << 8) | ((uint32_t)(r & 0xFF) << 16) |
                        ((uint32_t)((j == in.Jr[r] - 1 ? 1u : 0u) | 2u) << 24);
                uint32_t em = 0;
                for (int m = 0; m < in.M; ++m) {
                    const int pv = in.p[(size_t)k * in.M + m];
                    if (pv > 0) em |= 1u << m;
                    p[(size_t)k * MP + m] = (uint16_t)pv;
                    x[(size_t)k * MP + m] = in.x[(size_t)k * in.M + m];
                }
                elig[k] = em;
                uint32_t f4 = 0;
                for (int q = 0; q < in.elig_n[k] && q < 4; ++q) f4 |= (uint32_t)in.elig_list[(size_t)k * in.M + q] << (8 * q);
                first4[k] = f4;
            }
            jbeg += cnt;
        }
        if (b.grp) {
            // group kernels (fjsp_group.hip): one line of per-lane words behind the packed operation rows -- lane l:
            // len(machine l .kind_task_tuple) (the divisor of Machine.gap_ave, class_FJSSP.py:144-146) | first operation type of
            // job l << 8 | its J_r << 16 | (lanes 0, 1, 2: K, M, jobs) << 24; then the due date of job l (one job per kind: job = kind)
            uint32_t *hw = reinterpret_cast<uint32_t *>(ip(i, L.i_op) + 2048);
            int32_t *dj = reinterpret_cast<int32_t *>(ip(i, L.i_op) + 2048 + 64);
            for (int l = 0; l < 16; ++l) {
                uint32_t w = 0;
                if (l < in.M)
                    for (int k = 0; k < in.K; ++k) w += in.p[(size_t)k * in.M + l] > 0 ? 1u : 0u;
                if (l < in.R) { w |= (uint32_t)in.koff[l] << 8; w |= (uint32_t)in.Jr[l] << 16; dj[l] = due[l]; }
                w |= (uint32_t)(l == 0 ? in.K : (l == 1 ? in.M : (l == 2 ? nj : 0))) << 24;
                hw[l] = w;
            }
        }
        if (dyn) {
            uint16_t *pw = reinterpret_cast<uint16_t *>(ip(i, L.i_pw));
            int32_t *ipw = reinterpret_cast<int32_t *>(ip(i, L.i_ipw));
            uint16_t *bko = reinterpret_cast<uint16_t *>(ip(i, L.i_bkoff));
            int32_t *bk = reinterpret_cast<int32_t *>(ip(i, L.i_bk));
            for (int k = 0; k < in.K; ++k)
                for (int m = 0; m < in.M; ++m) pw[(size_t)k * MP + m] = (uint16_t)in.power[(size_t)k * in.M + m];
            int off = 0;
            for (int m = 0; m < (int)MP; ++m) {
                bko[m] = (uint16_t)off;
                if (m < in.M) { ipw[m] = in.idle_power[m]; off += in.bk_n[m]; }
            }
            bko[MP] = (uint16_t)off;
            for (int q = 0; q < 2 * off; ++q) bk[q] = in.bk[(size_t)q];
            reinterpret_cast<double *>(ip(i, L.i_ss))[0] = in.ddt;                         // observation[0] = self.DDT
        }
        if (variant == FJSP_VARIANT_MO_FJSSP_DISCRETES) {
            // MO_FJSSP_discretes.py:55-64 static_state_extract
            long ns = 0, js = 0;
            for (int r = 0; r < in.R; ++r) { ns += in.count[r]; js += in.Jr[r]; }
            const double N_ave = (double)ns / (double)in.R, J_ave = (double)js / (double)in.R;
            double a = 0.0, c2 = 0.0;
            for (int r = 0; r < in.R; ++r) a = a + std::pow((double)in.count[r] - N_ave, 2.0);
            for (int r = 0; r < in.R; ++r) c2 = c2 + std::pow((double)in.Jr[r] - J_ave, 2.0);
            double *ss = reinterpret_cast<double *>(ip(i, L.i_ss));
            ss[0] = in.ddt; ss[1] = (double)in.M; ss[2] = (double)in.R; ss[3] = N_ave;
            ss[4] = std::sqrt(a / (double)in.R); ss[5] = J_ave; ss[6] = std::sqrt(c2 / (double)in.R);
        }
        {   // fluid_completed_time = max_k Q_k / rate_k, rate_k summed over machine_rj_dict in FILE order
            // (class_FJSSP.py:276-278); the SO_SFJSP reward divides by it (SO_SFJSP.py:220)
            double best = 0.0;
            bool first_k = true;
            for (int k = 0; k < in.K; ++k) {
                double acc = 0.0;
                for (int q = 0; q < in.elig_n[k]; ++q) {
                    const int m = in.elig_list[(size_t)k * in.M + q];
                    acc = acc + in.x[(size_t)k * in.M + m] * (1.0 / (double)in.p[(size_t)k * in.M + m]);
                }
                int r_of_k = 0;
                while (in.koff[r_of_k + 1] <= k) ++r_of_k;
                const double v = (double)in.count[r_of_k] / acc;
                if (first_k || v > best) { best = v; first_k = false; }
            }
            reinterpret_cast<double *>(ip(i, L.i_ss))[7] = best;
        }
        // algorithmic HBM bytes of one env-step (DESIGN.md "bytes per env-step"):
        //   static per-k rows (kinfoB, elig, fmask u32; rate_sum, time_sum f64; see per_k below)    K * 28 ..
        //   job table read (due, jinfo, jst) + jst write-back                               njobs * 16
        //   machine lanes tend/mjob read + write                                            M * 16
        //   instance header + EnvScalars read + write                                       16 + 2 * 144
        //   column gather at k_sel (p u16, un/arr/rate f64) + un write                      M * 26 + 8
        //   actions in, state/reward/done out                                               2 + S*8 + 8 + 1
        // (per-k rows: kB, elig, fmask u32 + rate_sum, time_sum f64 = 28 B; + kA when a kind can have several jobs, + first4
        //  beyond 8 machines (CPython set order); + the 64-byte statistics row, read and written, in multi-job batches)
        //  single-job batches: + the assigned-machine byte; the column gather has no unprocessed entries there (p u16 + {arrival,
        //  rate} f64 = 18 B per machine, one byte written) against p + unprocessed + {arrival, rate} = 26 B and 8 B written)
        const double per_k = 28.0 + (b.single_job ? 1.0 : 4.0) + (b.MP > 8 ? 4.0 : 0.0) + (b.single_job ? 0.0 : 128.0);
        const double gather = b.single_job ? in.M * 18.0 + 1.0 : in.M * 26.0 + 8.0;
        bytes_acc += in.K * per_k + nj * 16.0 + in.M * 16.0 + 304.0 + gather + 2.0 + b.state_size * 8.0 + 9.0;
        if (dyn) bytes_acc += in.M * 14.0 + 2.0 * sizeof(DynScalars);    // power column, idle power, last-task ends r/w, DynScalars r/w
    }
    e->step_bytes = (int64_t)(bytes_acc / (double)NI + 0.5);

    {
        void *pi = nullptr, *pe = nullptr;
        if (!hip_ok(hipMalloc(&pi, islab.size()), "hipMalloc instance slab")) { delete e; return FJSP_E_HIP; }
        e->allocs.push_back(pi);
        if (!hip_ok(hipMalloc(&pe, N * L.e_stride), "hipMalloc env slab")) { fjsp_env_destroy(e); return FJSP_E_HIP; }
        e->allocs.push_back(pe);
        void *pd = nullptr;
        if (!hip_ok(hipMalloc(&pd, N + 16), "hipMalloc scratch")) { fjsp_env_destroy(e); return FJSP_E_HIP; }
        e->allocs.push_back(pd);
        e->d_done_scratch = reinterpret_cast<uint8_t *>(pd);
        if (b.mord) {
            void *pp = nullptr;
            if (!hip_ok(hipMalloc(&pp, (N + 1) * 4), "hipMalloc pending list") || !hip_ok(hipMemset(pp, 0, (N + 1) * 4), "hipMemset")) { fjsp_env_destroy(e); return FJSP_E_HIP; }
            e->allocs.push_back(pp);
            b.pending_count = reinterpret_cast<uint32_t *>(pp);
            // staging of the LP service, one slot per env (worst case: every env parks in the same launch)
            void *pin = nullptr, *px = nullptr;
            if (!hip_ok(hipMalloc(&pin, N * 2 * KP * 2), "hipMalloc LP inputs") || (e->allocs.push_back(pin), false) ||
                !hip_ok(hipMalloc(&px, N * KP * MP * 8), "hipMalloc LP solutions") || (e->allocs.push_back(px), false) ||
                !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_pending), (N + 1) * 4, hipHostMallocDefault), "hipHostMalloc") ||
                !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_lp_in), N * 2 * KP * 2, hipHostMallocDefault), "hipHostMalloc") ||
                !hip_ok(hipHostMalloc(reinterpret_cast<void **>(&e->h_lp_x), N * KP * MP * 8, hipHostMallocDefault), "hipHostMalloc")) {
                fjsp_env_destroy(e); return FJSP_E_HIP;
            }
            b.lp_in = reinterpret_cast<uint16_t *>(pin);
            b.lp_x = reinterpret_cast<double *>(px);
            // the LPs of order arrivals on the device when every tableau this batch can meet fits the LDS of a CU
            size_t lds_max = 0;
            for (size_t i = 0; i < NI; ++i) {
                const Instance &in = s->v[(size_t)first + i];
                int nx = 0;
                for (int v : in.p) nx += v > 0 ? 1 : 0;
                lds_max = std::max(lds_max, lp_device_lds_bytes(in.K, in.M, nx, in.R, b.MP));
                if (nx + 1 + (in.K + in.M + in.K - in.R) + 1 > lp_device_max_columns()) lds_max = (size_t)1 << 30;
            }
            // Which service: one LP takes the device ~0.3 ms (a 48-pivot tableau of the industrial instances; a host core
            // needs ~0.05 ms) but 256 of them run at once, so the device wins when arrivals come in bursts of hundreds --
            // measured (tools/bench_dynamic.py --instances industrial): 4096 envs 32.0 M env-steps/s against the host
            // service's 34.9 M, 32768 envs 63.1 M against 54.2 M.  Default: the device from 16384 environments on;
            // FJSP_LP_IMPL=device / host decides for itself (A/B runs, and the parity test of the two).
            const char *impl = getenv("FJSP_LP_IMPL");
            const bool want_device = impl ? strcmp(impl, "device") == 0 : b.N >= 16384;
            if (lds_max <= 156 * 1024 && want_device) {
                void *pe2 = nullptr, *ps2 = nullptr;
                if (!hip_ok(hipMalloc(&pe2, 8), "hipMalloc LP error word") || (e->allocs.push_back(pe2), false) ||
                    !hip_ok(hipMalloc(&ps2, 16), "hipMalloc LP counters") || (e->allocs.push_back(ps2), false) ||
                    !hip_ok(hipMemset(pe2, 0, 8), "hipMemset") || !hip_ok(hipMemset(ps2, 0, 16), "hipMemset")) { fjsp_env_destroy(e); return FJSP_E_HIP; }
                e->d_lp_err = reinterpret_cast<uint32_t *>(pe2);
                e->d_lp_solved = reinterpret_cast<unsigned long long *>(ps2);
                e->lp_device = true;
                e->lp_lds = lds_max;
            }
        }
        b.inst = reinterpret_cast<unsigned char *>(pi);
        b.envs = reinterpret_cast<unsigned char *>(pe);
        b.kenv = nullptr;
        { const char *kv = getenv("FJSP_GROUP_KENV"); b.kenv_first = (kv && atoi(kv) == 0) ? 0 : 1; }
        if (b.grp) {       // operation types of every environment's instance (fjsp_group.hip: large-batch kernels)
            std::vector<uint8_t> kq(N);
            for (size_t q = 0; q < N; ++q) kq[q] = (uint8_t)s->v[(size_t)first + (NI == N ? q : q % NI)].K;
            void *pk = nullptr;
            if (!hip_ok(hipMalloc(&pk, N), "hipMalloc K table") || (e->allocs.push_back(pk), false) ||
                !hip_ok(hipMemcpy(pk, kq.data(), N, hipMemcpyHostToDevice), "upload K table")) { fjsp_env_destroy(e); return FJSP_E_HIP; }
            b.kenv = reinterpret_cast<const uint8_t *>(pk);
        }
        // every env starts done so that step() before reset() is flagged, like the
        // reference's uninitialised object would fail
        std::vector<unsigned char> eslab(N * L.e_stride, 0);
        for (size_t q = 0; q < N; ++q) reinterpret_cast<EnvScalars *>(eslab.data() + q * L.e_stride)->done = 1;
        if (!hip_ok(hipMemcpy(pi, islab.data(), islab.size(), hipMemcpyHostToDevice), "upload instance slab") ||
            !hip_ok(hipMemcpy(pe, eslab.data(), eslab.size(), hipMemcpyHostToDevice), "upload env slab")) {
            fjsp_env_destroy(e); return FJSP_E_HIP;
        }
    }
    if (launch_fluid_tables(b, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) {
        set_error("fluid_tables_kernel launch failed");
        fjsp_env_destroy(e);
        return FJSP_E_HIP;
    }
    {
        // one reset of every env publishes each instance's reset observation (i_obs0, read by the autoreset path
        // of step_kernel); afterwards every env is marked done again so that step() before reset() is flagged,
        // like the reference's uninitialised object would fail
        std::vector<int32_t> ones(N, 1);
        if (launch_reset(b, nullptr, nullptr, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy2D(b.envs + offsetof(EnvScalars, done), L.e_stride, ones.data(), 4, 4, N, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("initial reset failed");
            fjsp_env_destroy(e);
            return FJSP_E_HIP;
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
        if (e->h_pending) (void)hipHostFree(e->h_pending);
        if (e->h_lp_in) (void)hipHostFree(e->h_lp_in);
        if (e->h_lp_x) (void)hipHostFree(e->h_lp_x);
    }
    delete e->workers;               // (joins: LPs still queued are solved first)
    if (e->ring) {
        DeviceGuard guard(e->device);
        for (int q = 0; q < e->ring_n; ++q) {
            AsyncBatch &a = e->ring[q];
            if (a.d_count) (void)hipFree(a.d_count);
            if (a.d_lp_in) (void)hipFree(a.d_lp_in);
            if (a.h_ids) (void)hipHostFree(a.h_ids);
            if (a.h_lp_in) (void)hipHostFree(a.h_lp_in);
            if (a.h_x) (void)hipHostFree(a.h_x);
            for (hipEvent_t ev : {a.ev_head, a.ev_tail, a.ev_up}) if (ev) (void)hipEventDestroy(ev);
        }
        if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
        if (e->ev_step) (void)hipEventDestroy(e->ev_step);
        delete[] e->ring;
    }
    delete e->pool;
    delete e;
}

int fjsp_env_num_envs(const fjsp_env *e) { return e ? e->b.N : 0; }
int fjsp_env_state_size(const fjsp_env *e) { return e ? e->b.state_size : 0; }
int fjsp_env_device(const fjsp_env *e) { return e ? e->device : -1; }
int64_t fjsp_env_step_bytes(const fjsp_env *e) { return e ? e->step_bytes : 0; }
int fjsp_env_kernel_family(const fjsp_env *e) { return e ? e->b.grp : 0; }
int64_t fjsp_env_lp_solves(const fjsp_env *e) {
    if (!e) return 0;
    int64_t n = e->lp_solves;
    if (e->lp_device) {        // (synchronises: the counter lives on the device)
        DeviceGuard guard(e->device);
        unsigned long long dev = 0;
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&dev, e->d_lp_solved, 8, hipMemcpyDeviceToHost) == hipSuccess) n += (int64_t)dev;
    }
    return n;
}
int fjsp_env_lp_on_device(const fjsp_env *e) { return (e && e->lp_device) ? 1 : 0; }
int64_t fjsp_env_lp_device_pivots(const fjsp_env *e) {
    if (!e || !e->lp_device) return 0;
    DeviceGuard guard(e->device);
    unsigned long long dev = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&dev, e->d_lp_solved + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)dev;
}
int fjsp_env_lp_device_solve(fjsp_env *e, int32_t env, const int32_t *Q, const int32_t *n_now, double *x) {
    if (!e || !Q || !n_now || !x || env < 0 || env >= e->b.N) { set_error("fjsp_env_lp_device_solve: bad arguments"); return FJSP_E_ARG; }
    if (!e->lp_device) { set_error("fjsp_env_lp_device_solve: this batch keeps the host LP service"); return FJSP_E_UNSUPPORTED; }
    DeviceGuard guard(e->device);
    const DevBatch &b = e->b;
    const Instance &in = e->src->v[(size_t)e->first + (size_t)(env % b.n_inst)];
    std::vector<uint16_t> lpq((size_t)2 * b.KP, 0);
    for (int k = 0; k < in.K; ++k) { lpq[(size_t)k] = (uint16_t)Q[k]; lpq[(size_t)b.KP + k] = (uint16_t)n_now[k]; }
    const uint32_t id = (uint32_t)env;
    // (slot 0 of the staging arrays; the batch must be idle: no parked environments)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b.lp_in, lpq.data(), lpq.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b.pending_count + 1, &id, 4, hipMemcpyHostToDevice));
    if (launch_lp_device(b, nullptr, 1, b.pending_count + 1, b.lp_in, b.lp_x, e->d_lp_err, nullptr, e->lp_lds, nullptr) != 0) {
        set_error("lp_device_kernel launch failed"); return FJSP_E_HIP;
    }
    HIP_TRY(hipDeviceSynchronize());
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, e->d_lp_err, 4, hipMemcpyDeviceToHost));
    if (err) { const uint32_t z = 0; (void)hipMemcpy(e->d_lp_err, &z, 4, hipMemcpyHostToDevice); set_error("fluid LP failed on the device (code " + std::to_string(err) + ")"); return FJSP_E_LP; }
    std::vector<double> xs((size_t)b.KP * b.MP);
    HIP_TRY(hipMemcpy(xs.data(), b.lp_x, xs.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < in.K; ++k)
        for (int m = 0; m < in.M; ++m) x[(size_t)k * in.M + m] = xs[(size_t)k * b.MP + m];
    return FJSP_OK;
}
int fjsp_env_set_lp_threads(fjsp_env *e, int32_t n_threads) {
    if (!e || n_threads < 0) { set_error("fjsp_env_set_lp_threads: bad arguments"); return FJSP_E_ARG; }
    e->lp_threads = n_threads;
    return FJSP_OK;
}

namespace {
// Multi-order batches: after a step launch, solve the fluid LP of every env that stopped at an order
// arrival (class_FJSSP.py:239 on the live state) with the host simplex and let arrival_kernel finish
// those steps.  Synchronises the stream: order arrivals make step() blocking for such batches.
int service_arrivals_impl(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done, int16_t *d_trace,
                          hipStream_t st);
// A failure inside the service (LP iteration limit, HIP error) leaves envs parked with no way to finish their
// step: the pending list is emptied, so a later launch cannot run its slots past the staging arrays, and the
// handle is marked failed: every later step / rollout returns FJSP_E_STATE until the batch is destroyed.
int service_arrivals(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done, int16_t *d_trace,
                     hipStream_t st) {
    const int rc = service_arrivals_impl(e, d_mo, d_state, d_reward, d_done, d_trace, st);
    if (rc != FJSP_OK) {
        const std::string why = fjsp_last_error();
        (void)hipMemsetAsync(e->b.pending_count, 0, 4, st);
        e->failed = true;
        set_error("order-arrival service failed (" + why + "); the batch is unusable: destroy it");
    }
    return rc;
}
int service_arrivals_impl(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done, int16_t *d_trace,
                          hipStream_t st) {
    const DevBatch &b = e->b;
    if (e->lp_device) {
        // the whole service on the stream: LP kernel (one workgroup per parked env, count read on the device), arrival_kernel,
        // pending list emptied -- no host round trip, fjsp_env_step stays asynchronous
        if (launch_lp_device(b, b.pending_count, 0, b.pending_count + 1, b.lp_in, b.lp_x, e->d_lp_err, e->d_lp_solved, e->lp_lds, st) != 0) {
            set_error("lp_device_kernel launch failed"); return FJSP_E_HIP;
        }
        if (launch_arrival(b, d_mo, 0, b.pending_count + 1, b.lp_x, d_state, d_reward, d_done, d_trace, st, nullptr, false, b.pending_count) != 0) {
            set_error("arrival_kernel launch failed"); return FJSP_E_HIP;
        }
        HIP_TRY(hipMemsetAsync(b.pending_count, 0, 4, st));
        return FJSP_OK;
    }
    // (the sync below also orders this call after the previous call's solution upload, so the pinned staging
    // buffers are free again)
    HIP_TRY(hipMemcpyAsync(e->h_pending, b.pending_count, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t n = e->h_pending[0];
    if (n == 0) return FJSP_OK;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP;
    // env ids and LP inputs of every parked env: two copies, whatever n is (step_kernel packed them by slot)
    HIP_TRY(hipMemcpyAsync(e->h_pending + 1, b.pending_count + 1, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(e->h_lp_in, b.lp_in, (size_t)n * 2 * KP * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // one LP per parked env, independent: spread over the host cores
    std::atomic<int> fail{0};
    std::string err;
    std::mutex err_mu;
    auto solve_one = [&](uint32_t q) {
        if (fail.load()) return;
        const int env = (int)e->h_pending[1 + q];
        const Instance &in = e->src->v[(size_t)e->first + (size_t)(env % b.n_inst)];
        const uint16_t *lpq = e->h_lp_in + (size_t)q * 2 * KP;
        std::vector<int> Q(in.K), now(in.K);
        for (int k = 0; k < in.K; ++k) { Q[k] = lpq[(size_t)k]; now[k] = lpq[KP + (size_t)k]; }
        std::vector<double> xk((size_t)in.K * in.M, 0.0);
        double obj = 0.0;
        const std::string ck = LpCache::key(env % b.n_inst, lpq, KP, in.K);
        if (!e->lp_cache.find(ck, xk)) {
            if (solve_fluid_lp(in.R, in.M, in.Jr.data(), in.p.data(), Q.data(), now.data(), xk.data(), &obj) != 0) {
                std::lock_guard<std::mutex> g(err_mu);
                if (fail.fetch_add(1) == 0) err = fjsp_last_error();     // thread-local message of this worker
                return;
            }
            e->lp_cache.put(ck, xk);
        }
        double *xin = e->h_lp_x + (size_t)q * KP * MP;
        std::fill(xin, xin + KP * MP, 0.0);
        for (int k = 0; k < in.K; ++k)
            for (int m = 0; m < in.M; ++m) xin[(size_t)k * MP + m] = xk[(size_t)k * in.M + m];
    };
    int n_threads = e->lp_threads > 0 ? e->lp_threads : std::min((int)std::thread::hardware_concurrency(), 16);
    if (n_threads <= 0) n_threads = 1;
    if (n_threads == 1 || n == 1) {
        for (uint32_t q = 0; q < n; ++q) solve_one(q);
    } else {
        if (e->pool && (int)e->pool->workers.size() != n_threads - 1) { delete e->pool; e->pool = nullptr; }
        if (!e->pool) e->pool = new LpPool(n_threads - 1);
        e->pool->run(n, solve_one);
    }
    if (fail.load()) { set_error(err); return FJSP_E_LP; }
    HIP_TRY(hipMemcpyAsync(b.lp_x, e->h_lp_x, (size_t)n * KP * MP * 8, hipMemcpyHostToDevice, st));
    if (launch_arrival(b, d_mo, (int)n, b.pending_count + 1, b.lp_x, d_state, d_reward, d_done, d_trace, st) != 0) { set_error("arrival_kernel launch failed"); return FJSP_E_HIP; }
    HIP_TRY(hipMemsetAsync(b.pending_count, 0, 4, st));
    e->lp_solves += n;
    return FJSP_OK;
}
}  // namespace

// ------------------------------------------------------------------ asynchronous arrival service
namespace {
constexpr int kAsyncRing = 32;         // batches in flight: a parked env waits for its LP (0.05 .. 0.8 ms) while calls come every ~0.03 ms
constexpr uint32_t kAsyncHead = 64;    // parked envs whose ids + LP inputs travel with the count (more: a second copy)

// pinned mirrors of a batch for `need` parked envs (grow-only: a typical launch parks a few dozen envs, a launch right
// after a synchronised reset can park all of them)
int async_reserve(fjsp_env *e, AsyncBatch &a, uint32_t need) {
    if (need <= a.cap) return FJSP_OK;
    const DevBatch &b = e->b;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP;
    uint32_t cap = std::max<uint32_t>(kAsyncHead, a.cap);
    while (cap < need) cap *= 2;
    cap = std::min<uint32_t>(cap, (uint32_t)b.N);
    uint32_t *ids = nullptr; uint16_t *in = nullptr; double *x = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ids), ((size_t)cap + 1) * 4, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&in), (size_t)cap * 2 * KP * 2, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&x), (size_t)cap * KP * MP * 8, hipHostMallocDefault));
    if (a.h_ids) {           // keep what the head copy already delivered
        std::memcpy(ids, a.h_ids, ((size_t)std::min(a.cap, cap) + 1) * 4);
        std::memcpy(in, a.h_lp_in, (size_t)std::min(a.cap, cap) * 2 * KP * 2);
        (void)hipHostFree(a.h_ids); (void)hipHostFree(a.h_lp_in); (void)hipHostFree(a.h_x);
    }
    a.h_ids = ids; a.h_lp_in = in; a.h_x = x; a.cap = cap;
    return FJSP_OK;
}

bool async_solve_one(fjsp_env *e, AsyncBatch *a, uint32_t q) {
    const DevBatch &b = e->b;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP;
    const int env = (int)a->h_ids[1 + q];
    const Instance &in = e->src->v[(size_t)e->first + (size_t)(env % b.n_inst)];
    const uint16_t *lpq = a->h_lp_in + (size_t)q * 2 * KP;
    std::vector<int> Q(in.K), now(in.K);
    for (int k = 0; k < in.K; ++k) { Q[k] = lpq[(size_t)k]; now[k] = lpq[KP + (size_t)k]; }
    std::vector<double> xk((size_t)in.K * in.M, 0.0);
    double obj = 0.0;
    const std::string ck = LpCache::key(env % b.n_inst, lpq, KP, in.K);
    if (!e->lp_cache.find(ck, xk)) {
        if (solve_fluid_lp(in.R, in.M, in.Jr.data(), in.p.data(), Q.data(), now.data(), xk.data(), &obj) != 0) {
            std::lock_guard<std::mutex> g(a->err_mu);
            a->err = fjsp_last_error();              // (thread-local message of this worker)
            return false;
        }
        e->lp_cache.put(ck, xk);
    }
    double *xin = a->h_x + (size_t)q * KP * MP;
    std::fill(xin, xin + KP * MP, 0.0);
    for (int k = 0; k < in.K; ++k)
        for (int m = 0; m < in.M; ++m) xin[(size_t)k * MP + m] = xk[(size_t)k * in.M + m];
    return true;
}

int async_setup(fjsp_env *e) {
    if (e->ring) return FJSP_OK;
    const DevBatch &b = e->b;
    const size_t N = (size_t)b.N, KP = (size_t)b.KP, MP = (size_t)b.MP;
    e->ring = new AsyncBatch[kAsyncRing];
    e->ring_n = kAsyncRing;
    for (int q = 0; q < kAsyncRing; ++q) {
        AsyncBatch &a = e->ring[q];
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.d_count), (N + 1) * 4));
        HIP_TRY(hipMemset(a.d_count, 0, (N + 1) * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.d_lp_in), N * 2 * KP * 2));
        const int rc = async_reserve(e, a, kAsyncHead);
        if (rc != FJSP_OK) return rc;
        HIP_TRY(hipEventCreateWithFlags(&a.ev_head, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&a.ev_tail, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&a.ev_up, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_step, hipEventDisableTiming));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_resume_ids), N * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_resume_x), N * KP * MP * 8));
    e->allocs.push_back(e->d_resume_ids);
    e->allocs.push_back(e->d_resume_x);
    // the threads the box gives (16 per GPU); more only oversubscribes a latency-critical wait
    int n_threads = e->lp_threads > 0 ? std::min(e->lp_threads, (int)std::thread::hardware_concurrency())
                                      : std::min((int)std::thread::hardware_concurrency(), 16);
    if (n_threads <= 0) n_threads = 1;
    e->workers = new LpWorkers(n_threads, [e](AsyncBatch *a, uint32_t q) { return async_solve_one(e, a, q); });
    return FJSP_OK;
}

bool async_idle(const fjsp_env *e) {
    if (!e->ring) return true;
    for (int q = 0; q < e->ring_n; ++q) if (e->ring[q].state != AsyncBatch::FREE) return false;
    return true;
}

// hand a batch's LPs to the worker threads
void async_submit(fjsp_env *e, AsyncBatch *a) {
    a->solved.store(0); a->bad.store(0); a->left.store((int)a->n);
    a->state = AsyncBatch::SOLVING;
    e->workers->submit(a);
}

// Advance every batch as far as it can go without waiting (block: with waiting, until the ring is empty).  Batches
// whose LPs are solved are uploaded and finished by arrival_kernel, which writes their outputs and ready = 1.
int async_progress(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done, uint8_t *d_ready, hipStream_t st,
                   bool block, bool mark_resumed) {
    const DevBatch &b = e->b;
    const size_t KP = (size_t)b.KP, MP = (size_t)b.MP;
    for (;;) {
        bool busy = false;
        for (int off = 0; off < e->ring_n; ++off) {
            AsyncBatch &a = e->ring[(e->ring_next + off) % e->ring_n];       // oldest first
            if (a.state == AsyncBatch::HEAD_COPY) {
                hipError_t q = block ? hipEventSynchronize(a.ev_head) : hipEventQuery(a.ev_head);
                if (q == hipErrorNotReady) { busy = true; continue; }
                HIP_TRY(q);
                a.n = a.h_ids[0];
                if (a.n > (uint32_t)b.N) a.n = (uint32_t)b.N;
                e->async_parked += a.n;
                if (a.n == 0) { a.state = AsyncBatch::FREE; continue; }
                if (a.n <= kAsyncHead) { async_submit(e, &a); busy = true; continue; }
                { const int rc = async_reserve(e, a, a.n); if (rc != FJSP_OK) return rc; }
                HIP_TRY(hipMemcpyAsync(a.h_ids + 1 + kAsyncHead, a.d_count + 1 + kAsyncHead, (size_t)(a.n - kAsyncHead) * 4, hipMemcpyDeviceToHost, e->copy_stream));
                HIP_TRY(hipMemcpyAsync(a.h_lp_in + (size_t)kAsyncHead * 2 * KP, a.d_lp_in + (size_t)kAsyncHead * 2 * KP,
                                       (size_t)(a.n - kAsyncHead) * 2 * KP * 2, hipMemcpyDeviceToHost, e->copy_stream));
                HIP_TRY(hipEventRecord(a.ev_tail, e->copy_stream));
                a.state = AsyncBatch::TAIL_COPY;
                busy = true;
            } else if (a.state == AsyncBatch::TAIL_COPY) {
                hipError_t q = block ? hipEventSynchronize(a.ev_tail) : hipEventQuery(a.ev_tail);
                if (q == hipErrorNotReady) { busy = true; continue; }
                HIP_TRY(q);
                async_submit(e, &a);
                busy = true;
            } else if (a.state == AsyncBatch::SOLVING || a.state == AsyncBatch::SOLVED) {
                int sv = a.solved.load();
                if (sv == 0 && block) {
                    while ((sv = a.solved.load()) == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
                if (sv == 0) { busy = true; continue; }
                if (sv < 0) { set_error("order-arrival LP failed: " + a.err); return FJSP_E_LP; }
                HIP_TRY(hipMemcpyAsync(e->d_resume_ids, a.h_ids + 1, (size_t)a.n * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(e->d_resume_x, a.h_x, (size_t)a.n * KP * MP * 8, hipMemcpyHostToDevice, st));
                if (launch_arrival(b, d_mo, (int)a.n, e->d_resume_ids, e->d_resume_x, d_state, d_reward, d_done, nullptr, st, d_ready,
                                   mark_resumed) != 0) { set_error("arrival_kernel launch failed"); return FJSP_E_HIP; }
                HIP_TRY(hipEventRecord(a.ev_up, st));
                e->lp_solves += a.n;
                e->async_parked -= a.n;
                a.state = AsyncBatch::UPLOADING;
                busy = true;
            } else if (a.state == AsyncBatch::UPLOADING) {
                hipError_t q = block ? hipEventSynchronize(a.ev_up) : hipEventQuery(a.ev_up);
                if (q == hipErrorNotReady) { busy = true; continue; }
                HIP_TRY(q);
                a.state = AsyncBatch::FREE;
            }
        }
        if (!block || !busy) return FJSP_OK;
    }
}
}  // namespace

int fjsp_env_step_async(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset, double *d_state, double *d_reward,
                        uint8_t *d_done, uint8_t *d_ready, void *stream) {
    if (!e || !d_actions || !d_ready) { set_error("fjsp_env_step_async: null argument"); return FJSP_E_ARG; }
    if (e->failed) { set_error("fjsp_env_step_async: the order-arrival service of this batch failed earlier; destroy the batch"); return FJSP_E_STATE; }
    if (reinterpret_cast<uintptr_t>(d_actions) & 1) { set_error("fjsp_env_step_async: d_actions must be 2-byte aligned"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    hipStream_t st = (hipStream_t)stream;
    if (!e->b.mord) {                               // nothing ever parks: the plain step, every env ready
        if (launch_step(e->b, d_actions, d_mo, autoreset ? 1 : 0, d_state, d_reward, d_done, nullptr, st) != 0) { set_error("step_kernel launch failed"); return FJSP_E_HIP; }
        HIP_TRY(hipMemsetAsync(d_ready, 1, (size_t)e->b.N, st));
        return FJSP_OK;
    }
    int rc = async_setup(e);
    if (rc != FJSP_OK) return rc;
    rc = async_progress(e, d_mo, d_state, d_reward, d_done, d_ready, st, false, true);
    if (rc != FJSP_OK) { e->failed = true; return rc; }
    // a free batch for whatever parks in this launch (none free: wait for the oldest ones)
    AsyncBatch *slot = nullptr;
    for (int attempt = 0; attempt < 2 && !slot; ++attempt) {
        for (int off = 0; off < e->ring_n; ++off) {
            AsyncBatch &a = e->ring[(e->ring_next + off) % e->ring_n];
            if (a.state == AsyncBatch::FREE) { slot = &a; e->ring_next = (int)((&a - e->ring) + 1) % e->ring_n; break; }
        }
        if (!slot) {
            rc = async_progress(e, d_mo, d_state, d_reward, d_done, d_ready, st, true, true);
            if (rc != FJSP_OK) { e->failed = true; return rc; }
        }
    }
    if (!slot) { set_error("fjsp_env_step_async: no free batch"); e->failed = true; return FJSP_E_STATE; }
    // From here on environments resumed by async_progress carry their solutions and the launch parks new ones into
    // `slot`: an error on the way leaves them without a path back, so every failing exit marks the batch unusable
    // (as service_arrivals does for the blocking service).
    auto launch_and_track = [&]() -> int {
        HIP_TRY(hipMemsetAsync(slot->d_count, 0, 4, st));
        DevBatch b2 = e->b;
        b2.pending_count = slot->d_count;
        b2.lp_in = slot->d_lp_in;
        if (launch_step(b2, d_actions, d_mo, autoreset ? 1 : 0, d_state, d_reward, d_done, nullptr, st, d_ready) != 0) {
            set_error("step_kernel launch failed"); return FJSP_E_HIP;
        }
        const size_t KP = (size_t)e->b.KP;
        const uint32_t head = std::min<uint32_t>(kAsyncHead, (uint32_t)e->b.N);
        HIP_TRY(hipEventRecord(e->ev_step, st));
        HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->ev_step, 0));
        HIP_TRY(hipMemcpyAsync(slot->h_ids, slot->d_count, (size_t)(1 + head) * 4, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipMemcpyAsync(slot->h_lp_in, slot->d_lp_in, (size_t)head * 2 * KP * 2, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipEventRecord(slot->ev_head, e->copy_stream));
        return FJSP_OK;
    };
    rc = launch_and_track();
    if (rc != FJSP_OK) { e->failed = true; return rc; }
    slot->state = AsyncBatch::HEAD_COPY;
    return FJSP_OK;
}

int fjsp_env_arrivals_flush(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done, uint8_t *d_ready, void *stream) {
    if (!e) { set_error("fjsp_env_arrivals_flush: null env"); return FJSP_E_ARG; }
    if (async_idle(e)) return FJSP_OK;
    DeviceGuard guard(e->device);
    const int rc = async_progress(e, d_mo, d_state, d_reward, d_done, d_ready, (hipStream_t)stream, true, false);
    if (rc != FJSP_OK) e->failed = true;
    return rc;
}

int64_t fjsp_env_parked(const fjsp_env *e) { return e ? e->async_parked : 0; }
int64_t fjsp_env_lp_cache_hits(fjsp_env *e) { if (!e) return 0; std::lock_guard<std::mutex> g(e->lp_cache.mu); return e->lp_cache.hits; }

int fjsp_env_reset(fjsp_env *e, const uint8_t *d_mask, double *d_state, void *stream) {
    if (!e) { set_error("fjsp_env_reset: null env"); return FJSP_E_ARG; }
    if (!async_idle(e)) { set_error("fjsp_env_reset: environments are parked in the asynchronous arrival service; call fjsp_env_arrivals_flush first"); return FJSP_E_STATE; }
    DeviceGuard guard(e->device);
    if (launch_reset(e->b, d_mask, d_state, (hipStream_t)stream) != 0) { set_error("reset_kernel launch failed"); return FJSP_E_HIP; }
    return FJSP_OK;
}

int fjsp_env_step_traced(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset, double *d_state,
                         double *d_reward, uint8_t *d_done, int16_t *d_trace_km, void *stream) {
    if (!e || !d_actions) { set_error("fjsp_env_step: null argument"); return FJSP_E_ARG; }
    if (e->failed) { set_error("fjsp_env_step: the order-arrival service of this batch failed earlier; destroy the batch"); return FJSP_E_STATE; }
    if (!async_idle(e)) { set_error("fjsp_env_step: environments are parked in the asynchronous arrival service; call fjsp_env_arrivals_flush first"); return FJSP_E_STATE; }
    if (reinterpret_cast<uintptr_t>(d_actions) & 1) { set_error("fjsp_env_step_traced: d_actions must be 2-byte aligned"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    if (launch_step(e->b, d_actions, d_mo, autoreset ? 1 : 0, d_state, d_reward, d_done, d_trace_km, (hipStream_t)stream) != 0) {
        set_error("step_kernel launch failed"); return FJSP_E_HIP;
    }
    if (e->b.mord) return service_arrivals(e, d_mo, d_state, d_reward, d_done, d_trace_km, (hipStream_t)stream);
    return FJSP_OK;
}

int fjsp_env_step(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset, double *d_state,
                  double *d_reward, uint8_t *d_done, void *stream) {
    return fjsp_env_step_traced(e, d_actions, d_mo, autoreset, d_state, d_reward, d_done, nullptr, stream);
}

int fjsp_env_rollout(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t T, int16_t *d_trace_km,
                     double *d_reward, double *d_state_last, void *stream) {
    if (!e || !d_actions || T <= 0) { set_error("fjsp_env_rollout: bad arguments"); return FJSP_E_ARG; }
    if (e->failed) { set_error("fjsp_env_rollout: the order-arrival service of this batch failed earlier; destroy the batch"); return FJSP_E_STATE; }
    if (!async_idle(e)) { set_error("fjsp_env_rollout: environments are parked in the asynchronous arrival service; call fjsp_env_arrivals_flush first"); return FJSP_E_STATE; }
    if (reinterpret_cast<uintptr_t>(d_actions) & 1) { set_error("fjsp_env_rollout: d_actions must be 2-byte aligned"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    hipStream_t st = (hipStream_t)stream;
    if (!e->b.mord && rollout_lds_bytes(e->b) <= 64 * 1024) {
        if (launch_rollout(e->b, d_actions, d_mo, T, d_trace_km, d_reward, d_state_last, st) != 0) {
            set_error("rollout_kernel launch failed"); return FJSP_E_HIP;
        }
        return FJSP_OK;
    }
    // instance too large for the LDS-resident fused kernel: T step launches on the same stream
    const size_t N = (size_t)e->b.N;
    for (int s2 = 0; s2 < T; ++s2) {
        if (launch_step(e->b, d_actions + (size_t)s2 * N * 2, d_mo, 2, d_state_last, d_reward ? d_reward + (size_t)s2 * N : nullptr,
                        e->d_done_scratch, d_trace_km ? d_trace_km + (size_t)s2 * N * 2 : nullptr, st) != 0) {
            set_error("step_kernel launch failed"); return FJSP_E_HIP;
        }
        if (e->b.mord) {
            const int rc = service_arrivals(e, d_mo, d_state_last, d_reward ? d_reward + (size_t)s2 * N : nullptr, e->d_done_scratch,
                                            d_trace_km ? d_trace_km + (size_t)s2 * N * 2 : nullptr, st);
            if (rc != FJSP_OK) return rc;
        }
    }
    return FJSP_OK;
}

namespace {
bool actor_ok(const fjsp_actor_params *a) {
    return a && a->w1 && a->b1 && a->w2 && a->b2 && a->w3 && a->b3 && a->hidden == kActorH && a->state_size > 0 &&
           a->state_size <= 32 && a->n_actions > 0 && a->n_actions <= kActorAP;
}
ActorParams actor_of(const fjsp_actor_params *a) {
    ActorParams p;
    p.w1 = a->w1; p.b1 = a->b1; p.w2 = a->w2; p.b2 = a->b2; p.w3 = a->w3; p.b3 = a->b3;
    p.S = a->state_size; p.H = a->hidden; p.A = a->n_actions;
    return p;
}
}  // namespace

int fjsp_actor_forward(const fjsp_actor_params *actor, const double *d_state, int32_t n, float *d_probs, void *stream) {
    if (!d_state || !d_probs || n <= 0) { set_error("fjsp_actor_forward: bad arguments"); return FJSP_E_ARG; }
    if (!actor_ok(actor)) { set_error("fjsp_actor_forward: the in-kernel actor is state_size (<= 32) -> 128 -> 128 -> n_actions (<= 32)"); return FJSP_E_UNSUPPORTED; }
    if (launch_actor_forward(actor_of(actor), d_state, n, d_probs, (hipStream_t)stream) != 0) { set_error("actor_forward_kernel launch failed"); return FJSP_E_HIP; }
    return FJSP_OK;
}

int fjsp_env_rollout_policy(fjsp_env *e, fjsp_rollout *buf, const fjsp_actor_params *actor, const float *d_epsilon,
                            const uint64_t *d_seed, int32_t pair_div, int32_t T, const double *d_mo, const double *d_state_in,
                            float *d_flat_actions, float *d_log_prob, double *d_state_last, void *stream) {
    if (!e || !buf || !d_epsilon || !d_seed || !d_state_in || !d_flat_actions || !d_log_prob || !d_state_last || T <= 0 || pair_div < 0) {
        set_error("fjsp_env_rollout_policy: bad arguments"); return FJSP_E_ARG;
    }
    if (e->failed) { set_error("fjsp_env_rollout_policy: the order-arrival service of this batch failed earlier; destroy the batch"); return FJSP_E_STATE; }
    if (buf->N != e->b.N || buf->S != e->b.state_size || buf->T < T || buf->device != e->device) {
        set_error("fjsp_env_rollout_policy: the rollout buffer does not match the batch (N, state_size, T, device)"); return FJSP_E_ARG;
    }
    if (!actor_ok(actor) || actor->state_size != e->b.state_size) {
        set_error("fjsp_env_rollout_policy: the in-kernel actor is state_size (<= 32) -> 128 -> 128 -> n_actions (<= 32)"); return FJSP_E_UNSUPPORTED;
    }
    if (e->b.mord || e->b.KC != 1 || policy_rollout_lds_bytes(e->b, actor->state_size) > 160 * 1024) {
        set_error("fjsp_env_rollout_policy: single-order batches of at most 64 operation types only"); return FJSP_E_UNSUPPORTED;
    }
    DeviceGuard guard(e->device);
    PolicyRolloutIO io;
    io.state_in = d_state_in; io.epsilon = d_epsilon; io.seed = d_seed; io.pair_div = pair_div;
    io.o_state = buf->states; io.o_actions = buf->actions; io.o_reward = buf->rewards; io.o_next = buf->next_states;
    io.o_done = buf->dones; io.o_valid = buf->valid; io.o_flat = d_flat_actions; io.o_logp = d_log_prob; io.state_last = d_state_last;
    if (launch_rollout_policy(e->b, actor_of(actor), io, d_mo, T, (hipStream_t)stream) != 0) {
        set_error("rollout_policy_kernel launch failed"); return FJSP_E_HIP;
    }
    buf->len = T;
    return FJSP_OK;
}

int fjsp_env_read(fjsp_env *e, int64_t *d_delay_time_sum, int32_t *d_makespan, int32_t *d_completion,
                  int32_t *d_step_time, int32_t *d_step_count, uint8_t *d_done, uint32_t *d_status, void *stream) {
    if (!e) { set_error("fjsp_env_read: null env"); return FJSP_E_ARG; }
    if (e->failed) { set_error("fjsp_env_read: the order-arrival service of this batch failed earlier; destroy the batch"); return FJSP_E_STATE; }
    if (!async_idle(e)) { set_error("fjsp_env_read: environments are parked in the asynchronous arrival service; call fjsp_env_arrivals_flush first"); return FJSP_E_STATE; }
    DeviceGuard guard(e->device);
    if (launch_read(e->b, d_delay_time_sum, d_makespan, d_completion, d_step_time, d_step_count, d_done, d_status,
                    (hipStream_t)stream) != 0) { set_error("read_kernel launch failed"); return FJSP_E_HIP; }
    if (e->lp_device) {        // a failed device LP surfaces here (read-back is where callers synchronise anyway)
        uint32_t err = 0;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(&err, e->d_lp_err, 4, hipMemcpyDeviceToHost));
        if (err) { e->failed = true; set_error("an order-arrival LP failed on the device (code " + std::to_string(err) + "); the batch is unusable: destroy it"); return FJSP_E_LP; }
    }
    return FJSP_OK;
}

int fjsp_env_machine_time_end(fjsp_env *e, int32_t *d_tend, int32_t m_stride, void *stream) {
    if (!e || !d_tend || m_stride < e->b.MP) { set_error("fjsp_env_machine_time_end: bad arguments"); return FJSP_E_ARG; }
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemcpy2DAsync(d_tend, (size_t)m_stride * 4, e->b.envs + e->b.L.e_tend, (size_t)e->b.L.e_stride,
                             (size_t)e->b.MP * 4, (size_t)e->b.N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FJSP_OK;
}

int fjsp_env_energy(fjsp_env *e, int64_t *d_energy, void *stream) {
    if (!e || !d_energy) { set_error("fjsp_env_energy: null argument"); return FJSP_E_ARG; }
    if (e->b.variant != FJSP_VARIANT_MO_DFJSP) { set_error("fjsp_env_energy: not a MO_DFJSP batch"); return FJSP_E_STATE; }
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemcpy2DAsync(d_energy, 8, e->b.envs + e->b.L.e_dyn, (size_t)e->b.L.e_stride, 8, (size_t)e->b.N,
                             hipMemcpyDeviceToDevice, (hipStream_t)stream));
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
    std::vector<double> col(MP * KP * 2), rs(KP), ts(KP);
    HIP_TRY(hipDeviceSynchronize());
    const unsigned char *rec = e->b.inst + (size_t)inst * e->b.L.i_stride;
    HIP_TRY(hipMemcpy(col.data(), rec + e->b.L.i_col, MP * KP * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rs.data(), rec + e->b.L.i_rsum, KP * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts.data(), rec + e->b.L.i_tsum, KP * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) {
        for (int m = 0; m < M; ++m) {
            if (h_rate) h_rate[(size_t)k * M + m] = col[((size_t)k * MP + m) * 2 + 1];
            if (h_arr) h_arr[(size_t)k * M + m] = col[((size_t)k * MP + m) * 2];
        }
        if (h_rate_sum) h_rate_sum[k] = rs[(size_t)k];
        if (h_time_sum) h_time_sum[k] = ts[(size_t)k];
    }
    return FJSP_OK;
}

}  // extern "C"
