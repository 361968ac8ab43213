// gfx950 (MI355X / CDNA4) kernels of the batched rule-dispatch FJSP environment.
//
// Execution model: ONE 64-lane WAVEFRONT PER ENVIRONMENT, 4 environments per
// 256-thread workgroup, no inter-wave communication at all.
//   * lane l of chunk c owns operation type k = 64c + l (the reference's
//     kind_task_tuple order): eligibility masks, fluid rates, per-k list
//     statistics live in that lane's registers;
//   * lanes 0..M-1 double as machine lanes (time_end, job on the machine);
//   * the job table (state word, due date, kind info) of the environment is
//     staged in a wave-private LDS slice; the fused rollout kernel also keeps
//     the machine x op "unprocessed" matrix there;
//   * static data comes from ONE record per instance and dynamic state from ONE
//     record per environment (fjsp_device.h), and every start-up load is issued
//     before the first wait, so a step costs two dependent memory round trips
//     (state in, column gather at the chosen operation) instead of a chain;
//   * next-event selection is a DPP min-reduction over the machine lanes,
//     availability sets are 64-bit ballots, a task rule's argmax/argmin is a
//     wave reduction of the key followed by the FIRST set bit of
//     ballot(member && key == extremum), a machine rule's walks its few
//     candidates in the list's (CPython set) order with readlane -- either way
//     the reference's "first extremum wins" tie-break (CPython max/min) is
//     kept -- and every float sum that feeds a decision or the observation is
//     accumulated serially in the reference's order (tree reductions would
//     break bit-exactness): the operands go to LDS rows and one lane per row
//     walks them.  Integer statistics use ballots / DPP row reductions.
//
// Variants (template parameter V): SO_FJSSP, SO_SFJSP, MO_FJSSP_discretes, an internal multi-order
// SO_FJSSP (order arrivals re-solve the fluid LP), and MO_DFJSP(_breakdown) on top of the latter.
//
// Reference restated (paths relative to the reference root):
//   environments/SO_FJSSP.py:51-76    reset            -> init_episode + observe
//   environments/SO_FJSSP.py:99-166   update_parameter -> compute_params (job-centric form)
//   environments/SO_FJSSP.py:168-265  step             -> decide / dispatch / advance_clock / observe
//   environments/SO_FJSSP.py:267-322  task_select, machine_select
//   environments/SO_FJSSP.py:78-97    state_extract    -> observe
//   environments/class_FJSSP.py:282-306 update_fluid_parameter -> fluid_tables_kernel, arrival_kernel
//   environments/SO_FJSSP.py:218-231    order arrival    -> order_arrive + host LP service (fjsp_env.hip) + arrival_kernel
//   environments/MO_DFJSP_breakdown.py:189-447 breakdown windows, energy, 12 x 10 rules, 15 observations, 4 rewards
// Compiled with -ffp-contract=off: a*b+c must round twice like CPython.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/fjsp_amd.h"
#include "fjsp_device.h"
#include "fjsp_common.h"
#include "fjsp_pyset.h"
#include "fjsp_policy.h"

#pragma clang fp contract(off)

namespace fjsp {

#ifndef FJSP_SHARED_TAIL
#define FJSP_SHARED_TAIL 1
#endif
constexpr uint32_t kAbsent = 0xFFu;      // next_stage of a job whose order has not arrived yet

// ------------------------------------------------------------------ diagnostics
// -DFJSP_STAMPS builds a DIAGNOSTIC library (never shipped, never benchmarked): lane 0 of
// every wave adds the s_memtime delta of each phase of a step to a global table.
#ifdef FJSP_STAMPS
__device__ unsigned long long fjsp_stamp_acc[16];
#define STAMP_FIELDS unsigned long long st[14]; unsigned long long st_t0;
#define STAMP_BEGIN(w) do { for (int _i = 0; _i < 14; ++_i) (w).st[_i] = 0; (w).st_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP(w, slot)                                                      \
    do {                                                                    \
        const unsigned long long _t1 = __builtin_amdgcn_s_memtime();            \
        (w).st[slot] += _t1 - (w).st_t0;                                    \
        (w).st_t0 = _t1;                                                    \
    } while (0)
#define STAMP_FLUSH(w)                                                                      \
    do {                                                                                    \
        if (__lane_id() == 0) {                                                             \
            for (int _i = 0; _i < 14; ++_i) atomicAdd(&fjsp_stamp_acc[_i], (w).st[_i]);     \
            atomicAdd(&fjsp_stamp_acc[15], 1ull);                                           \
        }                                                                                   \
    } while (0)
#else
#define STAMP_FIELDS
#define STAMP_BEGIN(w)
#define STAMP(w, slot)
#define STAMP_FLUSH(w)
#endif

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rlu(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double rld(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// lane `l` (a constant) of `vec` := the wave-uniform `sval` (v_writelane_b32; this clang has no builtin for it)
__device__ __forceinline__ int wlane(int sval, int l, int vec) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(__builtin_amdgcn_readfirstlane(sval)), "n"(l));
    return vec;
}
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// Hand-off of LDS data between lanes of one wave.  LDS operations of a wave execute
// in order, so only the compiler has to be kept from moving LDS accesses across the
// point; the fence is restricted to the LDS address space so it never drains the
// wave's outstanding global loads/stores (a plain wavefront fence costs a
// `s_waitcnt vmcnt(0)` = one HBM round trip every time).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// same, for data handed over through global memory (reset writing the unprocessed matrix)
__device__ __forceinline__ void wave_sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
extern __shared__ __attribute__((aligned(16))) unsigned char fjsp_lds[];

// RING = registers of the ring of the compiler-scheduled form (kernels whose waves each walk their own rows and
// hide the LDS behind one another).
template <int RING>
__device__ __forceinline__ double lds_chain_sum(const double *src, int n8) {
    if constexpr (RING == 8) return lds_chain_sum_ring8(src, n8);
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double acc = 0.0;
    double2 A[RING];
#pragma unroll
    for (int q = 0; q < RING; ++q) A[q] = s2[q];
    int i = 0;
    // whole turns of the ring: every register is refilled, unconditionally (no branches, no copies), as soon as
    // its two operands are consumed
    for (; i + 2 * RING <= n8; i += 2 * RING) {
#pragma unroll
        for (int q = 0; q < RING; ++q) {
            acc = acc + A[q].x; acc = acc + A[q].y;
            A[q] = s2[i / 2 + RING + q];
        }
    }
    // n8 is a multiple of 8: what is left is a multiple of 8 below 2 RING, already in the ring
#pragma unroll
    for (int h = 0; h < RING / 4; ++h) {
        if (i < n8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc = acc + A[4 * h + q].x; acc = acc + A[4 * h + q].y; }
            i += 8;
        }
    }
    return acc;
}

// sum over the 64 lanes; every lane must be active.  4 DPP steps give each
// 16-lane row its total (quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
// row_mirror), the 4 row totals are combined on the scalar unit.
__device__ __forceinline__ int wave_sum(int v) {
    v += DPP(v, 0xB1, 0);
    v += DPP(v, 0x4E, 0);
    v += DPP(v, 0x141, 0);
    v += DPP(v, 0x140, 0);
    return rl(v, 0) + rl(v, 16) + rl(v, 32) + rl(v, 48);
}
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, DPP(v, 0xB1, 0x7fffffff));
    v = min(v, DPP(v, 0x4E, 0x7fffffff));
    v = min(v, DPP(v, 0x141, 0x7fffffff));
    v = min(v, DPP(v, 0x140, 0x7fffffff));
    return min(min(rl(v, 0), rl(v, 16)), min(rl(v, 32), rl(v, 48)));
}
// a[c] for a wave-uniform chunk index, written as selects between VALUES: the optimiser turns some
// spellings of this into an indexed load, which pins the whole wave state in scratch (checked per chunk
// count with -Rpass-analysis=kernel-resource-usage: every kernel must report ScratchSize 0)
template <int KC, class T>
__device__ __forceinline__ T pick(const T (&a)[KC], int c) {
    if constexpr (KC == 1) return a[0];
    else if constexpr (KC == 2) { T r = a[0]; if (c == 1) r = a[1]; return r; }
    else {
        static_assert(KC == 4, "chunk counts are 1, 2 or 4");
        const T lo = (c & 1) ? a[1] : a[0], hi = (c & 1) ? a[3] : a[2];
        return (c & 2) ? hi : lo;
    }
}
// ------------------------------------------------------------------ wave state
template <int KC, int V>
struct W {
    // batch constants copied out of the kernel argument (keeping a pointer to the argument struct
    // makes the compiler spill it to scratch and re-load fields through memory)
    int KP, MP, JP, n_obs, n_static, state_size;
    bool single_job;         // DevBatch::single_job
    bool jreg;               // (compile-time single-job kernels of <= 64 operation types) the job words live in registers, see jsk
    uint32_t e_jst, e_tend, e_mjob, e_un, e_dyn, e_stats, e_asg;
    const double *sstate;
    double fluid_completed_time;
    int env, inst, lane;
    // uniform instance dims
    int K, M, njobs;
    uint32_t mmask;
    // uniform dynamic scalars (EnvScalars)
    int t, step_count, done, n_unassigned, completion, completion_last;
    uint32_t status, seq_ctr, rng_calls, busy;
    long long tard_done, delay_sum;
    uint64_t env_seed;
    int t_arr, next_order, pending, n_orders;      // order arrivals (multi-order batches)
    int obs_stale;                                 // obs_prev_l is not v(t-1) (EnvScalars::obs_stale bit 0)
    int stats_ok;                                  // the record's e_stats are current (EnvScalars::obs_stale bit 1 clear)
    long long energy, energy_last;                 // MO_DFJSP: energy_consumption(_last)
    const unsigned char *ir;
    // lane = operation type
    uint32_t kA[KC], kB[KC], elig[KC], fmask[KC], first4[KC];
    // jreg: state word and due date of THE job of this lane's kind (one job per kind: job index == kind index), kept per
    // lane instead of per job in LDS: every list walk, dispatch and release then stays inside the lane's registers
    uint32_t jsk[KC];
    int duek[KC];
    // single-job batches: the machine this lane's operation type was assigned to (0xFF: not yet) -- Layout::e_asg
    uint32_t asg[KC];
    int q0[KC], tot[KC];     // fluid_unprocessed_number_start; jobs of the kind that have arrived
    double rate_sum[KC], time_sum[KC];
    int nun[KC], cnt_a[KC], cnt_e[KC], max_a[KC], fifo_cnt[KC], head_job[KC], due_min[KC], tard[KC];
    double max_e[KC], sum_e[KC];
    // lane = machine
    int tend_m, mjob_m;
    int tlast_m, ipw_m;      // MO_DFJSP: time_end of the machine's last task (-1 = none), idle power
    // lane i < n_obs: previous observation
    double obs_prev_l;
    // wave-private LDS
    uint32_t *jstL;
    int32_t *dueL;
    double *scrL;   // 16 doubles: the observation being assembled (slot 15: mean of the machines' time_end)
    int32_t *hdrL;  // 16 ints: what the observation tail needs of this environment (TailHdr)
    double *frL, *grL, *tdL;   // serial-sum operands: finish_rate[KP], gap_rate[KP], time_end[KP] (zero padded)
    double *unp;    // unprocessed_rj matrix [KP][MP] (op-major): LDS slice (rollout) or the env record (step)
    // rows of this instance / env record
    // op-major matrices of the instance record: p_i[k*MP+m]; col_i[(k*MP+m)*2 + {0: arrival, 1: rate}]
    const uint16_t *p_i;
    const uint16_t *pw_i;    // MO_DFJSP: power_mrj_dict, same layout as p_i
    const double *col_i;
    unsigned char *er;
    STAMP_FIELDS
};

__host__ __device__ inline size_t lds_bytes_per_wave(int JP, int MP, int KP, bool un_lds) {
    // [obs 16][tail header 8][3 rows of KP + 2 doubles (16-byte skew: lds_chain_sum)][un][jst, due], then padded so that
    // consecutive slices start 64 bytes apart modulo the 256-byte bank period: the walker of step_kernel reads
    // the rows of four slices side by side
    const size_t raw = (size_t)(24 + 3 * (KP + 2)) * 8 + (un_lds ? (size_t)MP * KP * 8 : 0) + (size_t)JP * 8;
    return raw + (size_t)((64 + 256 - (int)(raw % 256)) % 256);
}

// Bind the wave to its records and bring the environment in.  All loads below are
// independent of each other (bounds come from the kernel arguments, not from the
// instance header), so they are in flight together: one memory round trip.
// SJ: 1 = the batch is known (at compile time) to have one job per kind, 0 = known not to, -1 = look at the batch
template <int KC, int V, int SJ = -1>
__device__ __forceinline__ void open_env(W<KC, V> &w, const DevBatch *b, int env, unsigned char *lds, bool un_lds,
                                         bool load_state, bool want_stats = false) {
    w.KP = kWave * KC;       // (KP = 64 KC by construction, fjsp_env.hip)
    w.MP = b->MP; w.JP = b->JP; w.n_obs = kNObs<V>; w.n_static = kNStatic<V>;
    w.state_size = kNStatic<V> + 2 * kNObs<V>;
    w.single_job = SJ < 0 ? (!is_mord_v<V> && b->single_job != 0) : (SJ != 0);
    w.jreg = SJ == 1 && KC == 1;
    // (offsets that follow from the padded sizes: FixedOffsets, fjsp_device.h -- no kernel-argument fetch)
    using FO = FixedOffsets;
    constexpr uint32_t KPc = kWave * KC;
    w.e_tend = FO::e_tend(); w.e_mjob = FO::e_mjob((uint32_t)b->MP); w.e_jst = FO::e_jst((uint32_t)b->MP);
    w.e_un = FO::e_un((uint32_t)b->MP, (uint32_t)b->JP);
    w.e_asg = FO::e_asg((uint32_t)b->MP, (uint32_t)b->JP, KPc, w.single_job);
    w.e_dyn = b->L.e_dyn; w.e_stats = b->L.e_stats;
    w.env = env;
    w.lane = (int)__lane_id();
    w.inst = b->n_inst == b->N ? env : env % b->n_inst;     // (one instance per environment: no division)
    const int JP = b->JP, KP = kWave * KC, MP = b->MP;
    const Layout &L = b->L;
    const unsigned char *ir = b->inst + (size_t)w.inst * L.i_stride;
    const uint32_t e_stride = is_mord_v<V> ? L.e_stride : FO::e_stride_plain((uint32_t)MP, (uint32_t)JP, KPc, w.single_job);
    unsigned char *er = b->envs + (size_t)env * e_stride;
    w.er = er;
    // LDS carve: [obs 16][tail header 16 x i32][fr KP][gr KP][td KP][un (optional)][jst][due]
    w.scrL = reinterpret_cast<double *>(lds);
    w.hdrL = reinterpret_cast<int32_t *>(w.scrL + 16);
    w.frL = w.scrL + 24; w.grL = w.frL + KP + 2; w.tdL = w.grL + KP + 2;
    unsigned char *q = reinterpret_cast<unsigned char *>(w.tdL + KP + 2);
    if (un_lds) { w.unp = reinterpret_cast<double *>(q); q += (size_t)MP * KP * 8; }
    else w.unp = reinterpret_cast<double *>(er + w.e_un);
    w.jstL = reinterpret_cast<uint32_t *>(q); q += (size_t)JP * 4;
    w.dueL = reinterpret_cast<int32_t *>(q);
    // ---- issue every load
    const InstHeader h = *reinterpret_cast<const InstHeader *>(ir);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = c * kWave + w.lane;
        w.kB[c] = reinterpret_cast<const uint32_t *>(ir + FO::i_kB(KPc))[k];
        // one job per kind: first job = kind index, one job (nothing to fetch); padding lanes have no job
        w.kA[c] = 0;
        if (!w.single_job) w.kA[c] = reinterpret_cast<const uint32_t *>(ir + FO::i_kA(KPc))[k];
        w.elig[c] = reinterpret_cast<const uint32_t *>(ir + FO::i_elig(KPc))[k];
        // the file order of the first four machines only matters to CPython's set order beyond 8 machines (fjsp_pyset.h)
        w.first4[c] = 0;
        if (MP > 8) w.first4[c] = reinterpret_cast<const uint32_t *>(ir + FO::i_f4(KPc))[k];
        w.asg[c] = 0xFFu;
        if (w.single_job && load_state) w.asg[c] = reinterpret_cast<const uint8_t *>(er + w.e_asg)[asg_pos((uint32_t)k)];
        if (is_mord_v<V> && load_state) {      // tables of the last LP of THIS environment
            w.fmask[c] = reinterpret_cast<const uint32_t *>(er + L.e_fmask)[k];
            w.rate_sum[c] = reinterpret_cast<const double *>(er + L.e_rsum)[k];
            w.time_sum[c] = reinterpret_cast<const double *>(er + L.e_tsum)[k];
            w.q0[c] = (int)reinterpret_cast<const uint32_t *>(er + L.e_q0)[k];
        } else {
            w.fmask[c] = reinterpret_cast<const uint32_t *>(ir + FO::i_fmask(KPc))[k];
            w.rate_sum[c] = reinterpret_cast<const double *>(ir + FO::i_rsum(KPc))[k];
            w.time_sum[c] = reinterpret_cast<const double *>(ir + FO::i_tsum(KPc))[k];
            w.q0[c] = 0;                     // = jobs of the kind, set below
        }
    }
    // update_parameter's statistics as the previous kernel left them (batches with several jobs per kind)
    uint4 st4[KC][4];
    const bool load_stats = load_state && want_stats && !w.single_job;      // (kernels that recompute them anyway pass want_stats = false)
    if (load_stats) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(er + L.e_stats) + (size_t)(c * kWave + w.lane) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) st4[c][q] = sp[q];
        }
    }
    const int jcap = b->jcap;
    int32_t due0 = 0;
    if (w.lane < jcap) due0 = reinterpret_cast<const int32_t *>(ir + FO::i_due(KPc))[w.lane];       // JP >= 64
    uint32_t jst0 = 0;
    int tend0 = 0, mjob0 = -1, tlast0 = -1, ipw0 = 0;
    double obs0 = 0.0;
    long long en0 = 0, enl0 = 0;
    if (V == kDyn && w.lane < MP) ipw0 = reinterpret_cast<const int32_t *>(ir + L.i_ipw)[w.lane];
    EnvScalars sc;      // uniform address: the compiler fetches it with scalar loads, no cross-lane traffic
    if (load_state) {
        if (w.lane < jcap) jst0 = reinterpret_cast<const uint32_t *>(er + w.e_jst)[w.lane];
        const EnvScalars *es = reinterpret_cast<const EnvScalars *>(er);
        sc.t = es->t; sc.step_count = es->step_count; sc.done = es->done; sc.n_unassigned = es->n_unassigned;
        sc.status = es->status; sc.seq_ctr = es->seq_ctr; sc.rng_calls = es->rng_calls; sc.busy = es->busy;
        sc.completion = es->completion; sc.completion_last = es->completion_last;
        sc.tard_done = es->tard_done; sc.delay_sum = es->delay_sum;
        sc.t_arr = es->t_arr; sc.next_order = es->next_order; sc.pending = es->pending; sc.obs_stale = es->obs_stale;
        if (w.lane < 10) obs0 = es->obs_prev[w.lane];
        if (w.lane < MP) {
            tend0 = reinterpret_cast<const int32_t *>(er + w.e_tend)[w.lane];
            mjob0 = reinterpret_cast<const int32_t *>(er + w.e_mjob)[w.lane];
        }
        if (V == kDyn) {
            const DynScalars *ds = reinterpret_cast<const DynScalars *>(er + L.e_dyn);
            en0 = ds->energy; enl0 = ds->energy_last;
            if (w.lane >= 10 && w.lane < 15) obs0 = ds->obs_hi[w.lane - 10];
            if (w.lane < MP) tlast0 = reinterpret_cast<const int32_t *>(er + L.e_dyn + sizeof(DynScalars))[w.lane];
        }
    }
    w.p_i = reinterpret_cast<const uint16_t *>(ir + L.i_p);
    w.pw_i = reinterpret_cast<const uint16_t *>(ir + (V == kDyn ? L.i_pw : L.i_p));
    w.col_i = reinterpret_cast<const double *>(is_mord_v<V> ? er + L.e_col : ir + L.i_col);
    w.ir = ir;
    w.t_arr = 0; w.next_order = 1; w.pending = 0; w.obs_stale = 0; w.stats_ok = 0;
    w.env_seed = b->rng_seed + (uint64_t)env * 1000003ULL;
    w.sstate = reinterpret_cast<const double *>(ir + L.i_ss);
    w.fluid_completed_time = w.sstate[7];
    // ---- consume
    w.K = uni(h.K); w.M = uni(h.M); w.njobs = uni(h.njobs);
    w.n_orders = is_mord_v<V> ? uni(h.R >> 16) : 1;       // InstHeader.R carries S in its high half for multi-order batches
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (w.single_job) w.kA[c] = ((w.kB[c] >> 24) & 2u) ? ((1u << 16) | ((w.kB[c] >> 16) & 0xFFu)) : 0u;
        w.tot[c] = (int)(w.kA[c] >> 16);
        if (!(is_mord_v<V> && load_state)) w.q0[c] = w.tot[c];
    }
    w.mmask = w.M >= 32 ? 0xFFFFFFFFu : ((1u << w.M) - 1u);
    w.ipw_m = ipw0; w.tlast_m = -1; w.energy = 0; w.energy_last = 0;
    if (w.jreg) {
        // lane k takes the words of job r(k) from lane r(k) (which fetched job r's words): one cross-lane move each
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int r = (int)((w.kB[c] >> 16) & 0xFFu) & 63;
            w.duek[c] = __shfl(due0, r, 64);
            w.jsk[c] = (uint32_t)__shfl((int)jst0, r, 64);
        }
    } else {
        if (w.lane < JP) w.dueL[w.lane] = due0;       // (row-kernel batches keep 16 job words: JP = 16)
        for (int n = kWave + w.lane; n < w.njobs; n += kWave) w.dueL[n] = reinterpret_cast<const int32_t *>(ir + FO::i_due(KPc))[n];
    }
    if (!load_state) return;
    if (!w.jreg) {
        if (w.lane < JP) w.jstL[w.lane] = jst0;
        for (int n = kWave + w.lane; n < w.njobs; n += kWave) w.jstL[n] = reinterpret_cast<const uint32_t *>(er + w.e_jst)[n];
    }
    w.t = uni(sc.t); w.step_count = uni(sc.step_count); w.done = uni(sc.done); w.n_unassigned = uni(sc.n_unassigned);
    w.status = uniu(sc.status); w.seq_ctr = uniu(sc.seq_ctr); w.rng_calls = uniu(sc.rng_calls); w.busy = uniu(sc.busy);
    w.completion = uni(sc.completion); w.completion_last = uni(sc.completion_last);
    w.tard_done = sc.tard_done; w.delay_sum = sc.delay_sum;
    w.t_arr = uni(sc.t_arr); w.next_order = uni((int)sc.next_order); w.pending = uni((int)sc.pending);
    w.obs_stale = uni((int)sc.obs_stale) & 1;
    w.stats_ok = !(uni((int)sc.obs_stale) & 2);
    w.obs_prev_l = obs0;
    w.tend_m = w.lane < w.M ? tend0 : 0;
    w.mjob_m = w.lane < w.M ? mjob0 : -1;
    if (V == kDyn) { w.tlast_m = w.lane < w.M ? tlast0 : -1; w.energy = en0; w.energy_last = enl0; }
    if (un_lds && !w.single_job) {
        const double *src = reinterpret_cast<const double *>(er + w.e_un);
        for (int i = w.lane; i < w.K * MP; i += kWave) w.unp[i] = src[i];
    }
    if (load_stats) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            w.nun[c] = (int)st4[c][0].x; w.cnt_a[c] = (int)st4[c][0].y; w.cnt_e[c] = (int)st4[c][0].z; w.max_a[c] = (int)st4[c][0].w;
            w.fifo_cnt[c] = (int)st4[c][1].x; w.head_job[c] = (int)st4[c][1].y; w.due_min[c] = (int)st4[c][1].z; w.tard[c] = (int)st4[c][1].w;
            w.max_e[c] = __hiloint2double((int)st4[c][2].y, (int)st4[c][2].x); w.sum_e[c] = __hiloint2double((int)st4[c][2].w, (int)st4[c][2].z);
            if (is_mord_v<V>) w.tot[c] = (int)st4[c][3].x;
        }
    }
    wave_sync();
}

// store_stats: also leave update_parameter's statistics in the record (what step_kernel starts from in batches with
// several jobs per kind); kernels that keep them to themselves mark the record's copy invalid instead.
template <int KC, int V>
__device__ __forceinline__ void store_dynamic(W<KC, V> &w, bool un_lds, bool store_stats = true) {
    unsigned char *er = w.er;
    wave_sync();
    if (w.lane == 0) {
        EnvScalars *es = reinterpret_cast<EnvScalars *>(er);
        es->t = w.t; es->step_count = w.step_count; es->done = w.done; es->n_unassigned = w.n_unassigned;
        es->status = w.status; es->seq_ctr = w.seq_ctr; es->rng_calls = w.rng_calls; es->busy = w.busy;
        es->completion = w.completion; es->completion_last = w.completion_last;
        es->tard_done = w.tard_done; es->delay_sum = w.delay_sum;
        es->t_arr = w.t_arr; es->next_order = (int16_t)w.next_order; es->pending = (int8_t)w.pending;
        es->obs_stale = (int8_t)((w.obs_stale & 1) | ((store_stats || w.single_job) ? 0 : 2));
    }
    if (w.lane < 10) reinterpret_cast<EnvScalars *>(er)->obs_prev[w.lane] = w.obs_prev_l;
    if (w.lane < w.M) {
        reinterpret_cast<int32_t *>(er + w.e_tend)[w.lane] = w.tend_m;
        reinterpret_cast<int32_t *>(er + w.e_mjob)[w.lane] = w.mjob_m;
    }
    if (V == kDyn) {
        DynScalars *ds = reinterpret_cast<DynScalars *>(er + w.e_dyn);
        if (w.lane == 0) { ds->energy = w.energy; ds->energy_last = w.energy_last; }
        if (w.lane >= 10 && w.lane < 15) ds->obs_hi[w.lane - 10] = w.obs_prev_l;
        if (w.lane < w.M) reinterpret_cast<int32_t *>(er + w.e_dyn + sizeof(DynScalars))[w.lane] = w.tlast_m;
    }
    if (w.jreg) {
#pragma unroll
        for (int c = 0; c < KC; ++c)       // the lane of a kind's first operation writes the kind's job word back
            if (((w.kB[c] >> 24) & 2u) && (w.kB[c] & 0xFFu) == 0) reinterpret_cast<uint32_t *>(er + w.e_jst)[(w.kB[c] >> 16) & 0xFFu] = w.jsk[c];
    } else {
        for (int n = w.lane; n < w.njobs; n += kWave) reinterpret_cast<uint32_t *>(er + w.e_jst)[n] = w.jstL[n];
    }
    if (!w.single_job && store_stats) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            uint4 *sp = reinterpret_cast<uint4 *>(er + w.e_stats) + (size_t)(c * kWave + w.lane) * 4;
            sp[0] = make_uint4((uint32_t)w.nun[c], (uint32_t)w.cnt_a[c], (uint32_t)w.cnt_e[c], (uint32_t)w.max_a[c]);
            sp[1] = make_uint4((uint32_t)w.fifo_cnt[c], (uint32_t)w.head_job[c], (uint32_t)w.due_min[c], (uint32_t)w.tard[c]);
            sp[2] = make_uint4((uint32_t)__double2loint(w.max_e[c]), (uint32_t)__double2hiint(w.max_e[c]),
                               (uint32_t)__double2loint(w.sum_e[c]), (uint32_t)__double2hiint(w.sum_e[c]));
            sp[3] = make_uint4((uint32_t)w.tot[c], 0u, 0u, 0u);
        }
    }
    if (un_lds && !w.single_job) {
        double *dst = reinterpret_cast<double *>(er + w.e_un);
        for (int i = w.lane; i < w.K * w.MP; i += kWave) dst[i] = w.unp[i];
    }
}

// ------------------------------------------------ update_parameter, job-centric
// SO_FJSSP.py:126-154 per operation type k=(r,j): the reference walks
// task_unprocessed_list[k] (jobs of kind r whose stage-j task is unassigned, in
// job-number order) with a positional index; here lane k walks the jobs of its
// kind and keeps the same index.  job_now_list[k] is recovered from the job
// words: jobs at stage j carrying a FIFO sequence; its head is the smallest
// sequence (append order, SO_FJSSP.py:215, class_FJSSP.py:225).
template <int KC, int V>
__device__ __forceinline__ void compute_params(W<KC, V> &w) {
    const int t = w.t;
    const double td = (double)t;
    // one job per kind (10x5, the Brandimarte sets): every list has at most one member, so the walk below is a
    // handful of selects -- no loop, no divergent branches
    if (w.single_job) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const uint32_t a = w.kA[c];
            const int jbeg = (int)(a & 0xFFFFu), j = (int)(w.kB[c] & 0xFFu);
            const bool has = (a >> 16) != 0;
            const uint32_t js = w.jreg ? w.jsk[c] : w.jstL[jbeg];
            const int d = w.jreg ? w.duek[c] : w.dueL[jbeg];
            const int nj = (int)(js & 0xFFu);
            const bool in = has && nj <= j;                             // the job's stage-j task is still unassigned
            const int da = t - d;                                       // :138
            const double est = td + w.time_sum[c];                      // :136,139 with task_index 0
            const double de = est - (double)d;
            const bool late = in && t > d;
            const bool waiting = in && nj == j && (js >> 8) != kNoSeq;
            w.nun[c] = in; w.cnt_a[c] = late; w.tard[c] = late ? da : 0;
            w.cnt_e[c] = in && est > (double)d;
            w.max_a[c] = in ? da : 0; w.max_e[c] = in ? de : 0.0; w.sum_e[c] = in ? de : 0.0;   // (0.0 + de == de)
            w.fifo_cnt[c] = waiting; w.head_job[c] = waiting ? jbeg : -1; w.due_min[c] = waiting ? d : 0x7fffffff;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const uint32_t a = w.kA[c];
        const int jbeg = (int)(a & 0xFFFFu), jcnt = (int)(a >> 16), j = (int)(w.kB[c] & 0xFFu);
        const double ts = w.time_sum[c];
        int idx = 0, cnt_a = 0, cnt_e = 0, max_a = 0, fifo = 0, head = -1, dmin = 0x7fffffff, tard = 0, arrived = 0;
        uint32_t hseq = 0xFFFFFFFFu;
        double max_e = 0.0, sum_e = 0.0;
        for (int n = jbeg; n < jbeg + jcnt; ++n) {
            const uint32_t js = w.jstL[n];
            const int d = w.dueL[n];            // fetched together with the state word: one LDS latency, not two
            const int nj = (int)(js & 0xFFu);
            if (is_mord_v<V> && nj != (int)kAbsent) arrived++;
            if (nj <= j) {
                const int da = t - d;                                   // :138
                const double est = td + ts * (double)(idx + 1);        // :136,139
                const double de = est - (double)d;
                if (t > d) { cnt_a++; tard += da; }                     // :134-135 (:120-122 for the last stage)
                if (est > (double)d) cnt_e++;                           // :136-137
                if (idx == 0) { max_a = da; max_e = de; }
                else { max_a = max(max_a, da); if (de > max_e) max_e = de; }
                sum_e = sum_e + de;                                     // :153 sum() left to right
                idx++;
                const uint32_t sq = js >> 8;
                if (nj == j && sq != kNoSeq) {
                    fifo++;
                    if (sq < hseq) { hseq = sq; head = n; }
                    dmin = min(dmin, d);                                // class_FJSSP.py:79-84 due_date_min
                }
            }
        }
        w.nun[c] = idx; w.cnt_a[c] = cnt_a; w.cnt_e[c] = cnt_e; w.max_a[c] = max_a; w.max_e[c] = max_e;
        w.fifo_cnt[c] = fifo; w.head_job[c] = head; w.due_min[c] = dmin; w.tard[c] = tard;
        w.sum_e[c] = sum_e;                                             // :153 urgency = sum_e / nun, formed on demand
        if (is_mord_v<V>) w.tot[c] = arrived;
    }
}

__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, DPP(v, 0xB1, (int)0x80000000));
    v = max(v, DPP(v, 0x4E, (int)0x80000000));
    v = max(v, DPP(v, 0x141, (int)0x80000000));
    v = max(v, DPP(v, 0x140, (int)0x80000000));
    return max(max(rl(v, 0), rl(v, 16)), max(rl(v, 32), rl(v, 48)));
}

// unsigned max over the 64 lanes (every lane active): four DPP steps inside the 16-lane rows, the four row results
// combined on the scalar unit
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)DPP((int)v, 0xB1, 0));
    v = max(v, (uint32_t)DPP((int)v, 0x4E, 0));
    v = max(v, (uint32_t)DPP((int)v, 0x141, 0));
    v = max(v, (uint32_t)DPP((int)v, 0x140, 0));
    return max(max(rlu(v, 0), rlu(v, 16)), max(rlu(v, 32), rlu(v, 48)));
}

// first-extremum argmax / argmin over the set bits of a ballot (the reference's max(list, key=) / min(list, key=)
// over a list in kind_task_tuple order: the first element attaining the extremum wins).  The extremum is a
// wave reduction, the winner the first set bit of ballot(member && key == extremum).
//
// f64 keys are reduced as ORDER-PRESERVING 64-bit integers (sign-magnitude -> biased: negative values
// complemented, the others with the sign bit set; -0.0 first folded onto +0.0, which compare equal in Python),
// high word first, then the low word among the lanes that tie on the high word: two 32-bit DPP reductions of
// one instruction per step, where the f64 form costs two moves, a canonicalisation and a v_max_f64 per step
// (f64 has no DPP).  The keys are finite (sums and differences of bounded counters): no NaN ordering to care for.
template <int KC>
__device__ __forceinline__ int argmax_f64(const uint64_t (&mask)[KC], const double (&key)[KC]) {
    const int lane = (int)__lane_id();
    bool in[KC];
    uint32_t hi[KC], lo[KC];
    uint32_t mhi = 0, mlo = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (KC == 1 && (mask[c] & (mask[c] - 1)) == 0) return mask[c] ? (int)__builtin_ctzll(mask[c]) : -1;
        in[c] = ((mask[c] >> lane) & 1ull) != 0;
        const double z = key[c] + 0.0;                                  // -0.0 -> +0.0
        const uint32_t zh = (uint32_t)__double2hiint(z), zl = (uint32_t)__double2loint(z);
        const bool neg = (zh >> 31) != 0;
        hi[c] = in[c] ? (neg ? ~zh : (zh | 0x80000000u)) : 0u;          // members are > 0: a non-member never wins
        lo[c] = neg ? ~zl : zl;
        if (mask[c]) mhi = max(mhi, wave_max_u32(hi[c]));
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
        if (mask[c]) mlo = max(mlo, wave_max_u32(hi[c] == mhi ? lo[c] : 0u));
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const uint64_t hit = __ballot(in[c] && hi[c] == mhi && lo[c] == mlo);
        if (hit) return c * kWave + (int)__builtin_ctzll(hit);
    }
    return -1;
}
template <int KC, bool MAXIMISE>
__device__ __forceinline__ int argext_i32(const uint64_t (&mask)[KC], const int (&key)[KC]) {
    const int lane = (int)__lane_id();
    bool in[KC];
    int ext = (int)0x80000000;              // of key (MAXIMISE) or of -key - 1 = ~key (minimise: order-reversing, no overflow)
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (KC == 1 && (mask[c] & (mask[c] - 1)) == 0) return mask[c] ? (int)__builtin_ctzll(mask[c]) : -1;
        in[c] = ((mask[c] >> lane) & 1ull) != 0;
        const int kv = MAXIMISE ? key[c] : ~key[c];
        if (mask[c]) ext = max(ext, wave_max_i32(in[c] ? kv : (int)0x80000000));
    }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const uint64_t hit = __ballot(in[c] && (MAXIMISE ? key[c] : ~key[c]) == ext);
        if (hit) return c * kWave + (int)__builtin_ctzll(hit);
    }
    return -1;
}
template <int KC>
__device__ __forceinline__ int popc_masks(const uint64_t (&mask)[KC]) {
    int n = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) n += __builtin_popcountll(mask[c]);
    return n;
}
template <int KC>
__device__ __forceinline__ int nth_bit(const uint64_t (&mask)[KC], int idx) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        uint64_t m = mask[c];
        const int n = __builtin_popcountll(m);
        if (idx < n) {
            for (int i = 0; i < idx; ++i) m &= m - 1;
            return c * kWave + __builtin_ctzll(m);
        }
        idx -= n;
    }
    return -1;
}
// random.choice replacement (fjsp_oracle.h): index into a list of length n
template <int KC, int V>
__device__ __forceinline__ int rng_choice(W<KC, V> &w, int n) {
    const uint64_t u = splitmix64(w.env_seed + (uint64_t)w.rng_calls);
    w.rng_calls++;
    return (int)(((u >> 32) * (uint64_t)n) >> 32);
}

template <int KC, int V>
__device__ __forceinline__ bool any_available(const W<KC, V> &w, uint32_t idle) {
    uint64_t any = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) any |= __ballot(w.fifo_cnt[c] > 0 && (w.elig[c] & idle) != 0);
    return any != 0;
}

// gap_time = step_time - order_arrive_time (SO_FJSSP.py:237); order_arrive_time is 0 with a single order
template <int KC, int V>
__device__ __forceinline__ double fluid_dt(const W<KC, V> &w) {
    return (double)(is_mord_v<V> ? w.t - w.t_arr : w.t);
}
// fluid_unprocessed_number (SO_FJSSP.py:239-240) is a pure function of the clock: Q0 - rate_sum * gap_time
template <int KC, int V>
__device__ __forceinline__ double fluid_q(const W<KC, V> &w, int c) {
    return (double)w.q0[c] - w.rate_sum[c] * fluid_dt(w);
}

// SO_FJSSP.py:267-298 task_select.  Returns k or -1 (status set).
template <int KC, int V>
__device__ __forceinline__ int task_select(W<KC, V> &w, int a0, uint32_t idle) {
    uint64_t av[KC], fav[KC];
    uint64_t anyav = 0, anyfav = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const bool has = w.fifo_cnt[c] > 0;
        av[c] = __ballot(has && (w.elig[c] & idle) != 0);
        fav[c] = __ballot(has && (w.fmask[c] & idle) != 0);
        anyav |= av[c]; anyfav |= fav[c];
    }
    if (!anyav) { w.status |= FJSP_ST_NO_EVENT; return -1; }
    if (V == FJSP_VARIANT_SO_SFJSP) {                 // SO_SFJSP.py:169-188
        if (a0 == 0) {                                        // rule 1: argmax Tasks.gap
            double gap[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) gap[c] = (double)w.nun[c] - fluid_q(w, c);
            return anyfav ? argmax_f64<KC>(fav, gap) : argmax_f64<KC>(av, gap);
        }
        if (a0 == 1 || a0 == 2) {                             // rules 2, 3: argmin time_min[_fluid]_rj (:234-244)
            const bool fluid = a0 == 1 && anyfav;
            int tmin[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const uint32_t cand = idle & (fluid ? w.fmask[c] : w.elig[c]);
                int best = 0x7fffffff;
                for (int m = 0; m < w.M; ++m) {
                    const int pv = w.p_i[(c * kWave + w.lane) * w.MP + m];
                    if (((cand >> m) & 1u) && pv < best) best = pv;
                }
                tmin[c] = best;
            }
            return fluid ? argext_i32<KC, false>(fav, tmin) : argext_i32<KC, false>(av, tmin);
        }
        if (a0 == 3) return nth_bit<KC>(av, rng_choice(w, popc_masks<KC>(av)));   // rule 4
        w.status |= FJSP_ST_BAD_TASK_RULE;
        return -1;
    }
    if (V == kDyn && a0 >= 5) {                       // MO_DFJSP_breakdown.py:357-381, rules 6..12
        if (a0 == 5) return argext_i32<KC, false>(av, w.due_min);               // rule 6: argmin due date over available
        if (a0 <= 9) {                                // rules 7/8: argmin energy_min[_fluid]_rj, 9/10: argmin time_min[_fluid]_rj
            const bool energy = a0 <= 7;
            const bool fluid = (a0 == 6 || a0 == 8) && anyfav;
            int vmin[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const uint32_t cand = idle & (fluid ? w.fmask[c] : w.elig[c]);
                int best = 0x7fffffff;
                for (int m = 0; m < w.M; ++m) {
                    const int o = (c * kWave + w.lane) * w.MP + m;
                    const int pv = w.p_i[o];
                    const int v = energy ? pv * (int)w.pw_i[o] : pv;             // energy_mrj_dict class_MODFJSP.py:178
                    if (((cand >> m) & 1u) && v < best) best = v;
                }
                vmin[c] = best;
            }
            return fluid ? argext_i32<KC, false>(fav, vmin) : argext_i32<KC, false>(av, vmin);
        }
        if (a0 == 10 && anyfav) return nth_bit<KC>(fav, rng_choice(w, popc_masks<KC>(fav)));   // rule 11
        if (a0 == 10 || a0 == 11) return nth_bit<KC>(av, rng_choice(w, popc_masks<KC>(av)));   // rules 11, 12
        w.status |= FJSP_ST_BAD_TASK_RULE;
        return -1;
    }
    // kind_task_delivery_urgency (:153) = sum(estimated delays) / len(list); only rules 1, 2, 4 read it
    double urg[KC];
    if (a0 == 0 || a0 == 1 || a0 == 3) {
        // (one job per kind: a list of one, sum / 1.0 == sum; lanes outside the candidate masks are never read)
#pragma unroll
        for (int c = 0; c < KC; ++c) urg[c] = w.single_job ? w.sum_e[c] : w.sum_e[c] / (double)w.nun[c];
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) urg[c] = 0.0;
    }
    switch (a0) {
    case 0: {   // rule 1 :269-273
        uint64_t de[KC]; uint64_t any = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) { de[c] = av[c] & __ballot(w.cnt_e[c] > 0); any |= de[c]; }
        return any ? argmax_f64<KC>(de, w.max_e) : argmax_f64<KC>(av, urg);
    }
    case 1: {   // rule 2 :274-278
        uint64_t da[KC]; uint64_t any = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) { da[c] = av[c] & __ballot(w.cnt_a[c] > 0); any |= da[c]; }
        return any ? argext_i32<KC, true>(da, w.max_a) : argmax_f64<KC>(av, urg);
    }
    case 2: {   // rule 3 :279-283, Tasks.gap class_FJSSP.py:70-72
        double gap[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) gap[c] = (double)w.nun[c] - fluid_q(w, c);
        return anyfav ? argmax_f64<KC>(fav, gap) : argmax_f64<KC>(av, gap);
    }
    case 3:     // rule 4 :284-288
        return anyfav ? argmax_f64<KC>(fav, urg) : argmax_f64<KC>(av, urg);
    case 4:     // rule 5 :289-293
        return anyfav ? argext_i32<KC, false>(fav, w.due_min) : argext_i32<KC, false>(av, w.due_min);
    case 5:     // rule 6 :294-295
        return nth_bit<KC>(av, rng_choice(w, popc_masks<KC>(av)));
    default:
        w.status |= FJSP_ST_BAD_TASK_RULE;      // MyError :297
        return -1;
    }
}

// Machine.gap_ave (class_FJSSP.py:144-146) of up to three machines at once: a strictly sequential sum over
// kind_task_tuple order of unprocessed - fluid_unprocessed of the machine's operation types, divided by
// (n + 1e-18).  The gap rows of the three machines (ineligible entries +0.0, an exact identity of the
// running sum) go to the three LDS operand rows and lanes 0..2 walk one row each, so the chains cost one
// LDS read + one add per element instead of a ballot/readlane walk per machine.  Machine ids < 0 are
// skipped.  Returns, in lane q < 3, the gap_ave of machine mq.
template <int KC, int V, int RING>
__device__ __forceinline__ double gap_ave3(const W<KC, V> &w, int m0, int m1, int m2) {
    const double dt = fluid_dt(w);
    const int n8 = (w.K + 7) & ~7;
    // (offset arithmetic, not a select between the pointer fields of `w`: that would pin `w` in scratch)
    const double *src = w.frL + (w.lane == 1 ? w.KP + 2 : (w.lane == 2 ? 2 * (w.KP + 2) : 0));
    int cnt_q[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int m = q == 0 ? m0 : (q == 1 ? m1 : m2);
        double *row = q == 0 ? w.frL : (q == 1 ? w.grL : w.tdL);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * kWave + w.lane;
            // (r, j) is one of machine m's operation types iff m is in machine_rj_dict[(r, j)]: the eligibility mask
            // this lane already holds (no look-up of the processing time)
            const bool on_m = m >= 0 && ((w.elig[c] >> (m & 31)) & 1u) != 0;
            double g = 0.0;
            // (uniform: the machine ids are wave-uniform)
            if (on_m) {
                const int o = k * w.MP + m;
                const double2 ar = *reinterpret_cast<const double2 *>(w.col_i + 2 * o);
                // (single-job batches: unprocessed = arrival, less one where this type was assigned to m)
                const double un_km = w.single_job ? (w.asg[c] == (uint32_t)m ? ar.x - 1.0 : ar.x) : w.unp[o];
                g = un_km - (ar.x - dt * ar.y);
            }
            row[k] = g;
            cnt_q[q] += __builtin_popcountll(__ballot(on_m));
        }
    }
    wave_sync();
    const double sm = lds_chain_sum<RING>(src, n8);
    wave_sync();
    const int n = w.lane == 0 ? cnt_q[0] : (w.lane == 1 ? cnt_q[1] : cnt_q[2]);
    if (V == kDyn) return sm / (double)n;             // class_MODFJSP.py:158-159 has no epsilon
    return sm / ((double)n + 1e-18);
}

// SO_FJSSP.py:300-322 machine_select.  Candidate lists are visited in CPython's
// list(set & set) order (fjsp_pyset.h): ascending for M <= 8, not always beyond.
template <int KC, int V, int RING>
__device__ __forceinline__ int machine_select(W<KC, V> &w, int a1, int k_sel, uint32_t idle, int *p_sel, double *un_sel,
                                              int *en_sel) {
    const int cs = k_sel >> 6, ls = k_sel & 63;
    const uint32_t elig_s = rlu(pick<KC>(w.elig, cs), ls), fm_s = rlu(pick<KC>(w.fmask, cs), ls);
    CandList sel, fsel;
    if (w.MP <= 8) {
        // machine ids < 8: every CPython set involved iterates in ascending order (fjsp_pyset.h), the lists are masks
        sel.mask = idle & elig_s; sel.n = __builtin_popcount(sel.mask); sel.packed = 0; sel.asc = true;
        fsel.mask = idle & fm_s; fsel.n = __builtin_popcount(fsel.mask); fsel.packed = 0; fsel.asc = true;
    } else {
        const uint32_t first4 = rlu(pick<KC>(w.first4, cs), ls);
        sel = pyset_and(idle, elig_s, first4, false);              // machine_selectable_list :302
        fsel = pyset_and(idle, fm_s, 0u, true);                    // fluid_machine_selectable_list :303
    }
    if (sel.n == 0) { w.status |= FJSP_ST_NO_EVENT; return -1; }
    // lane m: gap_rj_dict[m][k_sel] (class_FJSSP.py:137-142) and p[m][k_sel]
    double g = 0.0, un = 0.0;
    int pm = 0, en = 0;
    if (w.lane < w.M && ((sel.mask >> w.lane) & 1u)) {
        // op-major layout: the column of k_sel is MP contiguous entries per array (a handful of cache lines)
        const int o = k_sel * w.MP + w.lane;
        pm = w.p_i[o];
        const double2 ar = *reinterpret_cast<const double2 *>(w.col_i + 2 * o);
        // (single-job batches: k_sel has not been dispatched yet -- it is available -- so its unprocessed is its arrival)
        un = w.single_job ? ar.x : w.unp[o];
        g = un - (ar.x - fluid_dt(w) * ar.y);
        if (V == kDyn) en = pm * (int)w.pw_i[o];                 // energy_mrj_dict class_MODFJSP.py:178
    }
    constexpr int base = 0;
    auto visit = [&](const CandList &l, auto &&f) __attribute__((always_inline)) {
        if (l.asc) { uint32_t m = l.mask; while (m) { f((int)__builtin_ctz(m)); m &= m - 1; } }
        else for (int i = 0; i < l.n; ++i) f((int)((l.packed >> (8 * i)) & 0xFFu));
    };
    // a candidate list in ascending order (always the case for M <= 8) is a lane mask: the rule is then the same
    // wave reduction + first-set-bit ballot as the task rules; the CPython-ordered short lists are walked.
    // `at` = the lane that holds machine 0's key.
    auto lane_argmax_f64 = [&](const CandList &l, double key, int at) __attribute__((always_inline)) {
        if (l.n == 1) return (int)__builtin_ctz(l.mask);
        if (l.asc) {
            const uint64_t m64[1] = {(uint64_t)l.mask << at};
            const double k1[1] = {key};
            return argmax_f64<1>(m64, k1) - at;
        }
        int best = -1; double bv = 0.0;
        visit(l, [&](int m) __attribute__((always_inline)) { const double v = rld(key, at + m); if (best < 0 || v > bv) { bv = v; best = m; } });
        return best;
    };
    auto lane_argmin_i32 = [&](const CandList &l, int key, int at) __attribute__((always_inline)) {
        if (l.n == 1) return (int)__builtin_ctz(l.mask);
        if (l.asc) {
            const uint64_t m64[1] = {(uint64_t)l.mask << at};
            const int k1[1] = {key};
            return argext_i32<1, false>(m64, k1) - at;
        }
        int best = -1, bv = 0;
        visit(l, [&](int m) __attribute__((always_inline)) { const int v = rl(key, at + m); if (best < 0 || v < bv) { bv = v; best = m; } });
        return best;
    };
    auto argmax_gap = [&](const CandList &l) __attribute__((always_inline)) { return lane_argmax_f64(l, g, base); };
    auto argmin_p = [&](const CandList &l) __attribute__((always_inline)) { return lane_argmin_i32(l, pm, base); };
    auto argmax_gave = [&](const CandList &l) __attribute__((always_inline)) {
        if (l.n == 1) return (int)__builtin_ctz(l.mask);
        // gap_ave of every candidate, three per pass, parked in the candidate's machine lane
        double gave_m = 0.0;
        uint32_t rest = l.mask;
        while (rest) {
            int mq[3] = {-1, -1, -1};
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (rest) { mq[q] = (int)__builtin_ctz(rest); rest &= rest - 1; }
            const double v = gap_ave3<KC, V, RING>(w, mq[0], mq[1], mq[2]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (mq[q] >= 0 && w.lane == mq[q]) gave_m = rld(v, q);
        }
        return lane_argmax_f64(l, gave_m, 0);
    };
    auto argmin_en = [&](const CandList &l) __attribute__((always_inline)) { return lane_argmin_i32(l, en, base); };
    auto argmin_ipw = [&](const CandList &l) __attribute__((always_inline)) { return lane_argmin_i32(l, w.ipw_m, 0); };
    int m_sel;
    if (V == kDyn) {
        const CandList &fl = fsel.n ? fsel : sel;
        switch (a1) {                                             // MO_DFJSP_breakdown.py:384-428
        case 0: m_sel = argmax_gap(fl); break;
        case 1: m_sel = argmin_p(fl); break;
        case 2: m_sel = argmin_p(sel); break;
        case 3: m_sel = argmax_gave(fl); break;
        case 4: m_sel = argmin_en(fl); break;                     // rule 5: least processing energy, fluid first
        case 5: m_sel = argmin_en(sel); break;
        case 6: m_sel = argmin_ipw(fl); break;                    // rule 7: least idle power, fluid first
        case 7: m_sel = argmin_ipw(sel); break;
        case 8: m_sel = cand_at(fl, rng_choice(w, fl.n)); break;
        case 9: m_sel = cand_at(sel, rng_choice(w, sel.n)); break;
        default:
            w.status |= FJSP_ST_BAD_MACHINE_RULE;
            return -1;
        }
    } else if (V == FJSP_VARIANT_SO_SFJSP) {
        switch (a1) {                                             // SO_SFJSP.py:190-214
        case 0: m_sel = argmax_gap(fsel.n ? fsel : sel); break;
        case 1: m_sel = argmin_p(fsel.n ? fsel : sel); break;
        case 2: m_sel = argmin_p(sel); break;
        case 3: m_sel = argmax_gave(fsel.n ? fsel : sel); break;
        case 4: m_sel = cand_at(sel, rng_choice(w, sel.n)); break;
        default:
            w.status |= FJSP_ST_BAD_MACHINE_RULE;
            return -1;
        }
    } else if (V == FJSP_VARIANT_MO_FJSSP_DISCRETES) {
        switch (a1) {                                             // MO_FJSSP_discretes.py:209-230
        case 0: m_sel = fsel.n ? argmax_gap(fsel) : argmin_p(sel); break;       // rule 1 :213-217
        case 1: m_sel = argmax_gave(fsel.n ? fsel : sel); break;                // rule 2 :218-222
        case 2: m_sel = argmax_gap(fsel.n ? fsel : sel); break;                 // rule 3 :223-227
        default:
            w.status |= FJSP_ST_BAD_MACHINE_RULE;                 // MyError :229
            return -1;
        }
    } else {
        switch (a1) {
        case 0: m_sel = argmax_gap(fsel.n ? fsel : sel); break;       // rule 1 :304-308
        case 1: m_sel = argmax_gap(sel); break;                       // rule 2 :309-310
        case 2: m_sel = argmin_p(sel); break;                         // rule 3 :311-312
        case 3: m_sel = argmax_gave(fsel.n ? fsel : sel); break;      // rule 4 :313-317
        case 4: m_sel = cand_at(sel, rng_choice(w, sel.n)); break;    // rule 5 :318-319
        default:
            w.status |= FJSP_ST_BAD_MACHINE_RULE;                     // MyError :321
            return -1;
        }
    }
    *p_sel = rl(pm, base + m_sel);
    *un_sel = rld(un, base + m_sel);
    if (V == kDyn) *en_sel = rl(en, base + m_sel);
    return m_sel;
}

// reset_object_add(order s) without its LP (class_FJSSP.py:205-237): the jobs of order s enter the stage-0
// FIFOs in (kind, job number) order, and the LP inputs Q[k] = len(task_unprocessed_list), n_now[k] =
// len(job_now_list) are written to the env record for the host service.  Only the LAST arrival of a step
// matters for the tables: nothing inside the event loop reads the fluid solution, and each LP resets the
// previous one's state.  Returns the number of jobs that arrived.
template <int KC, int V>
__device__ __forceinline__ int order_arrive(W<KC, V> &w, const DevBatch *b, int s_ord) {
    const Layout &L = b->L;
    const uint16_t *ocnt = reinterpret_cast<const uint16_t *>(w.ir + L.i_ocnt);
    const int RP = b->RP;
    int added = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const uint32_t kb = w.kB[c];
        const bool first_stage = ((kb >> 24) & 2u) && (kb & 0xFFu) == 0;     // one lane per kind
        const int r = (int)((kb >> 16) & 0xFFu);
        int cnt = 0, nstart = 0, before = 0;
        if (first_stage) {
            cnt = ocnt[s_ord * RP + r];
            for (int q = 0; q < s_ord; ++q) nstart += ocnt[q * RP + r];      // Kind.number_start, :212
            for (int q = 0; q < r; ++q) before += ocnt[s_ord * RP + q];      // jobs of earlier kinds come first
            const int jbeg = (int)(w.kA[c] & 0xFFFFu);
            for (int n = 0; n < cnt; ++n)                                    // :216-225
                w.jstL[jbeg + nstart + n] = jst_pack(w.seq_ctr + (uint32_t)(before + n), 0u);
            w.fifo_cnt[c] += cnt;
        }
        added += wave_sum(cnt);
    }
    w.seq_ctr += (uint32_t)added;
    w.n_unassigned += added;
    wave_sync();
    // LP inputs (:234-237): unprocessed tasks per operation type, jobs waiting per operation type
    uint16_t *lpq = reinterpret_cast<uint16_t *>(w.er + L.e_lpq);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const uint32_t a = w.kA[c];
        const int jbeg = (int)(a & 0xFFFFu), jcnt = (int)(a >> 16), j = (int)(w.kB[c] & 0xFFu);
        int q = 0;
        for (int n = jbeg; n < jbeg + jcnt; ++n) q += ((int)(w.jstL[n] & 0xFFu) <= j) ? 1 : 0;   // absent jobs carry 0xFF
        const int k = c * kWave + w.lane;
        if (k < w.K) { lpq[k] = (uint16_t)q; lpq[w.KP + k] = (uint16_t)w.fifo_cnt[c]; }
    }
    return added;
}

// SO_FJSSP.py:176-250: dispatch the FIFO head of k_sel on m_sel, then advance
// the clock until some operation type is available again (or the episode ends).
template <int KC, int V>
__device__ __forceinline__ void dispatch_and_advance(W<KC, V> &w, const DevBatch *b, int k_sel, int m_sel, int pm,
                                                     double un_sel, int en_sel) {
    const int cs = k_sel >> 6, ls = k_sel & 63;
    const int job = rl(pick<KC>(w.head_job, cs), ls);                       // :176 job_now_list[0]
    const uint32_t kb = rlu(pick<KC>(w.kB, cs), ls);
    const int Jr = (int)((kb >> 8) & 0xFFu);
    int time_end = w.t + pm;                                                 // :184
    int machine_end = time_end;
    if (V == kDyn) {
        // MO_DFJSP_breakdown.py:204-231: the windows of m_sel in file order; a window that covers the start or
        // begins inside the task stretches the task, one that begins exactly at its end only delays the machine
        const uint16_t *bko = reinterpret_cast<const uint16_t *>(w.ir + b->L.i_bkoff);
        const int32_t *bk = reinterpret_cast<const int32_t *>(w.ir + b->L.i_bk);
        const int q1 = bko[m_sel + 1];
        for (int q = bko[m_sel]; q < q1; ++q) {
            const int bs = bk[2 * q], be = bk[2 * q + 1];
            if (bs <= w.t && w.t < be) { time_end += be - w.t; machine_end = time_end; }
            else if (w.t < bs && bs < time_end) { time_end += be - bs; machine_end = time_end; }
            else if (bs == time_end) machine_end += be - bs;
            else if (bs > time_end) break;
        }
        // :253-256 processing energy, then the idle energy since the machine's previous task ended
        w.energy += (long long)en_sel;
        const int tl = rl(w.tlast_m, m_sel);
        if (tl >= 0) w.energy += (long long)(w.t - tl) * (long long)rl(w.ipw_m, m_sel);
        if (w.lane == m_sel) w.tlast_m = time_end;
    }
    const int nj = (int)(kb & 0xFFu) + 1;       // the FIFO head of (r, j) is at stage j; it moves to j + 1
    if (w.jreg) {
        const uint32_t r_sel = (kb >> 16) & 0xFFu;
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (((w.kB[c] >> 16) & 0xFFu) == r_sel) w.jsk[c] = jst_pack(kNoSeq, (uint32_t)nj);      // :186-191
    }
    if (w.single_job) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (c == cs && w.lane == ls) w.asg[c] = (uint32_t)m_sel;         // :198 unprocessed_rj_dict[m][(r, j)] -= 1
    }
    if (w.lane == 0) {
        if (!w.jreg) w.jstL[job] = jst_pack(kNoSeq, (uint32_t)nj);           // :186-191
        if (w.single_job) reinterpret_cast<uint8_t *>(w.er + w.e_asg)[asg_pos((uint32_t)k_sel)] = (uint8_t)m_sel;
        else w.unp[k_sel * w.MP + m_sel] = un_sel - 1.0;                     // :198
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
        if (c == cs && w.lane == ls) w.fifo_cnt[c]--;
    // machine lane: time_end, and job | (k of the job's next stage + 1) << 16 so that the release at the
    // completion event (:209-215) needs no look-up (0 in the high half = the job has no further stage)
    if (w.lane == m_sel) { w.tend_m = machine_end; w.mjob_m = job | ((nj == Jr ? 0 : k_sel + 2) << 16); }   // :194-197
    w.busy |= 1u << m_sel;
    if (time_end > w.completion) w.completion = time_end;
    if (nj == Jr) {                                                          // :200-202
        w.n_unassigned--;
        const int late = time_end - (w.jreg ? rl(pick<KC>(w.duek, cs), ls) : w.dueL[job]);
        w.tard_done += late > 0 ? late : 0;
    }
    uint32_t idle = ~w.busy & w.mmask;
    while (!any_available<KC, V>(w, idle)) {                                    // :204
        const int cand = (w.lane < w.M && w.tend_m > w.t) ? w.tend_m : 0x7fffffff;
        const int tn = wave_min(cand);                                       // :205-207 next event
        if (tn == 0x7fffffff) { w.status |= FJSP_ST_NO_EVENT; break; }
        w.t = tn;
        uint64_t fin = __ballot(w.lane < w.M && w.tend_m == tn);            // :209-215, ascending m
        while (fin) {
            const int m = __builtin_ctzll(fin);
            fin &= fin - 1;
            const int mj = rl(w.mjob_m, m);
            const int jb = mj & 0xFFFF, kk = (mj >> 16) - 1;
            if (kk >= 0) {
                const int n2 = (int)(rlu(pick<KC>(w.kB, kk >> 6), kk & 63) & 0xFFu);   // stage index of kk
                if (w.jreg) {
#pragma unroll
                    for (int c = 0; c < KC; ++c)
                        if ((int)((w.kB[c] >> 16) & 0xFFu) == jb) w.jsk[c] = jst_pack(w.seq_ctr, (uint32_t)n2);
                } else if (w.lane == 0) {
                    w.jstL[jb] = jst_pack(w.seq_ctr, (uint32_t)n2);
                }
                w.seq_ctr++;
#pragma unroll
                for (int c = 0; c < KC; ++c)
                    if (c == (kk >> 6) && w.lane == (kk & 63)) w.fifo_cnt[c]++;
            }
        }
        if (is_mord_v<V> && w.next_order < w.n_orders) {                     // :218-231 order arrival
            const int t_next = reinterpret_cast<const int32_t *>(w.ir + b->L.i_oarr)[w.next_order];
            const bool due_now = t_next <= w.t, idle_jump = !due_now && w.n_unassigned == 0;
            if (due_now || idle_jump) {
                order_arrive<KC, V>(w, b, w.next_order);
                w.next_order++;
                w.t_arr = t_next;
                if (idle_jump) w.t = t_next;                                 // :231 the clock jumps to the arrival
                w.pending = 1;
            }
        }
        w.busy &= ~(uint32_t)__ballot(w.lane < w.M && w.tend_m <= w.t);     // :233-235
        idle = ~w.busy & w.mmask;
        if (w.n_unassigned == 0 && !(is_mord_v<V> && w.next_order < w.n_orders)) { w.done = 1; break; }   // :247-250
    }
    wave_sync();
}

// ------------------------------------------------------------------ observation
// SO_FJSSP.py:78-97 state_extract (+ the ratios of update_parameter :156-165), in three parts:
//
//   observe_prepare   (every environment's own wave, lane-parallel) integer statistics, the operand rows of the
//                     three mean / population-std pairs (finish_rate, gap_rate, machine time_end) and a small
//                     header, all into the wave's LDS slice;
//   observe_tail      the strictly sequential part: two passes of left-to-right f64 sums over the rows (:86-95),
//                     the divisions and square roots.  One lane walks one row, so the whole tail of an
//                     environment occupies 8 lanes (3 chains + the 4 delay ratios) -- and costs a full wave's
//                     instruction stream all the same.  In step_kernel ONE wave of the workgroup (the walker)
//                     therefore runs the tails of all four environments of the workgroup side by side (lane =
//                     8 * slot + role), between workgroup barriers: the stream is issued once instead of four
//                     times.  Everywhere else (reset, fused rollout, arrival) each wave runs its own tail;
//   the caller        (emit_state) picks the finished observation up from the LDS scratch row.
//
// The tail lanes write their results straight to the observation's final positions (ObsPos), the mean of the
// machines' time_end -- needed for its deviations, not part of the observation -- to slot 15.
enum { H_K = 0, H_M, H_TASKS, H_JOBS, H_DELAY_A, H_DELAY_E, H_JOB_A, H_JOB_E, H_DONE, H_N8 };

template <bool SHARED>
__device__ __forceinline__ void tail_sync() {
    if constexpr (SHARED) {
        // workgroup barrier with LDS-only fences (a plain __syncthreads() also drains the global traffic)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
        wave_sync();
    }
}

// Integer statistics + operand rows + tail header.  Needs compute_params() at the current clock.  Returns
// tard_unproc (delay_time_sum_unprocessed, :110-122); frv / grv keep this lane's finish_rate / gap_rate for the
// deviations of the second pass.
template <int KC, int V>
__device__ __forceinline__ long long observe_prepare(W<KC, V> &w, bool stats_only, double (&frv)[KC], double (&grv)[KC]) {
    const int K = w.K, M = w.M;
    const bool single_job = w.single_job;      // every operation type has exactly one job (10x5, Mk01..10): counts are 0 or 1
    // ---- integer statistics (order-free)
    int task_number = 0, delay_a = 0, delay_e = 0, job_a = 0, job_e = 0;
    const int job_number = w.n_unassigned;
    long long tard_unproc;
    if (single_job && w.t < (1 << 22)) {
        // 0/1 counts: ballots + scalar popcounts; the tardiness of the (at most 64 KC) late last-stage
        // operations fits one 32-bit DPP reduction
        int tl = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const bool last = ((w.kB[c] >> 24) & 1u) != 0;
            task_number += __builtin_popcountll(__ballot(w.nun[c] > 0));
            delay_a += __builtin_popcountll(__ballot(w.cnt_a[c] > 0));
            delay_e += __builtin_popcountll(__ballot(w.cnt_e[c] > 0));
            job_a += __builtin_popcountll(__ballot(last && w.cnt_a[c] > 0));
            job_e += __builtin_popcountll(__ballot(last && w.cnt_e[c] > 0));
            tl += last ? w.tard[c] : 0;
        }
        tard_unproc = (long long)wave_sum(tl);
    } else {
        // packed DPP reductions (totals < 65536, checked at create)
        uint32_t nun_s = 0, a_s = 0, e_s = 0, ja_s = 0, je_s = 0;
        long long tu = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const bool last = ((w.kB[c] >> 24) & 1u) != 0;
            nun_s += (uint32_t)w.nun[c]; a_s += (uint32_t)w.cnt_a[c]; e_s += (uint32_t)w.cnt_e[c];
            if (last) { ja_s += (uint32_t)w.cnt_a[c]; je_s += (uint32_t)w.cnt_e[c]; tu += w.tard[c]; }
        }
        const uint32_t r1 = (uint32_t)wave_sum((int)(nun_s | (a_s << 16)));
        const uint32_t r2 = (uint32_t)wave_sum((int)(e_s | (ja_s << 16)));
        const uint32_t r3 = (uint32_t)wave_sum((int)(je_s | ((uint32_t)(tu >> 24) << 16)));
        const uint32_t r4 = (uint32_t)wave_sum((int)(tu & 0xFFFFFF));
        task_number = (int)(r1 & 0xFFFFu); delay_a = (int)(r1 >> 16);
        delay_e = (int)(r2 & 0xFFFFu); job_a = (int)(r2 >> 16);
        job_e = (int)(r3 & 0xFFFFu);
        tard_unproc = (long long)r4 + ((long long)(r3 >> 16) << 24);
    }
    // a rollout that never hands a state back (rule sweeps read makespan / tardiness / energy only) needs the
    // tardiness of the unfinished jobs for the reward and nothing else of the observation
    if (stats_only) return V == FJSP_VARIANT_SO_SFJSP ? 0 : tard_unproc;
    // ---- operand rows, zero padded to KP entries (+0.0 is an exact identity of a running sum from +0.0)
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int tot = w.tot[c];
        const int k = c * kWave + w.lane;
        const bool valid = k < K;
        const double fnum = (double)(tot - w.nun[c]), gnum = (double)w.nun[c] - fluid_q(w, c);
        if (single_job) { frv[c] = valid ? fnum : 0.0; grv[c] = valid ? gnum : 0.0; }     // x / 1.0 == x
        else {
            frv[c] = valid ? fnum / (double)tot : 0.0;              // finish_rate class_FJSSP.py:74-76
            grv[c] = valid ? gnum / (double)w.q0[c] : 0.0;          // gap_rate    class_FJSSP.py:66-68
        }
        w.frL[k] = frv[c]; w.grL[k] = grv[c];
        // lane 2 of the tail walks the machines' time_end: an exact integer sum, so the f64 walk equals
        // sum(int) / M (:384-385)
        w.tdL[k] = (c == 0 && w.lane < M) ? (double)w.tend_m : 0.0;
    }
    // ---- header for the tail lanes: one lane-indexed store
    int hv = 0;
    hv = wlane(K, H_K, hv); hv = wlane(M, H_M, hv);
    hv = wlane(task_number, H_TASKS, hv); hv = wlane(job_number, H_JOBS, hv);
    hv = wlane(delay_a, H_DELAY_A, hv); hv = wlane(delay_e, H_DELAY_E, hv);
    hv = wlane(job_a, H_JOB_A, hv); hv = wlane(job_e, H_JOB_E, hv);
    hv = wlane(w.done, H_DONE, hv);
    hv = wlane((max(K, M) + 7) & ~7, H_N8, hv);   // (a shop can have more machines than operation types)
    if (w.lane < 16) w.hdrL[w.lane] = hv;
    return tard_unproc;
}

// One pass of the tail for up to four environments whose LDS slices start at slot0 + e * stride (e < nslots;
// nslots == 1: the wave's own slice).  Lane L serves slot (L >> 3) & 3 in role L & 7: roles 0..2 walk the
// finish_rate / gap_rate / time_end rows, roles 4..7 form the delay ratios of :156-165.  First pass: the row
// sums become the three means (and the ratios are formed -- one division instruction for all of them).  (The second
// pass of the observation -- the standard deviations -- no longer walks: observe_deviations.)
template <int V, int RING>
__device__ __forceinline__ void tail_pass(unsigned char *slot0, uint32_t stride, int nslots, int KP) {
    const int lane = (int)__lane_id();
    const int e = (lane >> 3) & 3, r = lane & 7;
    const bool live = e < nslots;
    unsigned char *sl = slot0 + (live ? (uint32_t)e * stride : 0u);
    double *scr = reinterpret_cast<double *>(sl);
    const int32_t *hdr = reinterpret_cast<const int32_t *>(sl + 16 * 8);
    const double *row = reinterpret_cast<const double *>(sl + 24 * 8) + (r < 3 ? r * (KP + 2) : 0);
    // every header word this lane needs, fetched before the walk (the walk is hand-scheduled asm: nothing moves across it)
    const int n8_l = live ? hdr[H_N8] : 0;
    const int len_i = hdr[r == 2 ? H_M : H_K];
    const int rnum = hdr[r >= 4 ? r : H_DELAY_A];                           // roles 4..7: delay_a, delay_e, job_a, job_e
    const int rden = hdr[r < 6 ? H_TASKS : H_JOBS];
    const int done = hdr[H_DONE];
    // the longest row of the (up to four) slots bounds everybody's walk; rows are zero padded to KP
    int n8 = max(max(rl(n8_l, 0), rl(n8_l, 8)), max(rl(n8_l, 16), rl(n8_l, 24)));
    n8 = max(n8, 8);
    const double csum = lds_chain_sum<RING>(row, n8);
    using P = ObsPos<V>;
    const double ave = (r < 4 ? csum : (double)rnum) / (r < 4 ? (double)len_i : (double)rden);
    const double out = (r >= 4 && done) ? 0.0 : ave;                   // the ratios are 0 once the episode is over (:156-165)
    const int pos = r == 0 ? P::ave0 : (r == 1 ? P::ave1 : (r == 2 ? P::ave2 : P::ratio0 + r - 4));
    if (live && r != 3) scr[pos] = out;
}

// Second pass of the observation by the environment's own wave: the squared deviations from the means the first pass left
// in the scratch row (math.pow(d, 2), :86-95), their sums by the fixed tree of fjsp_common.h (row_tree_sum_f64: P_l over
// k = l, l + 16, l + 32, ... left to right, then the 16-lane butterfly -- the same tree the row kernels of fjsp_group.hip
// use), and the three population standard deviations, written to their places in the scratch row.  Lane (rho, l) of chunk c
// holds k = 64 c + 16 rho + l: every lane fetches the four rows' values of its column with ds_bpermute and adds them in row
// order, chunk after chunk, so all four rows hold the same P_l.
template <int KC, int V>
__device__ __forceinline__ void observe_deviations(W<KC, V> &w, const double (&frv)[KC], const double (&grv)[KC]) {
    using P = ObsPos<V>;
    const double ave_fr = w.scrL[P::ave0], ave_gr = w.scrL[P::ave1], ave_td = w.scrL[P::ave2];
    const int l = w.lane & 15;
    double p_fr = 0.0, p_gr = 0.0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = c * kWave + w.lane;
        const bool valid = k < w.K;
        const double d1 = frv[c] - ave_fr, d2 = grv[c] - ave_gr;
        const double v1 = valid ? d1 * d1 : 0.0, v2 = valid ? d2 * d2 : 0.0;
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            p_fr = p_fr + __shfl(v1, 16 * rho + l, 64);
            p_gr = p_gr + __shfl(v2, 16 * rho + l, 64);
        }
    }
    const double d3 = (double)w.tend_m - ave_td;
    const double v3 = w.lane < w.M ? d3 * d3 : 0.0;                   // (at most 32 machines: rows 0 and 1)
    double p_td = 0.0;
    p_td = p_td + __shfl(v3, l, 64);
    p_td = p_td + __shfl(v3, 16 + l, 64);
    const double s_fr = row_tree_sum_f64(p_fr), s_gr = row_tree_sum_f64(p_gr), s_td = row_tree_sum_f64(p_td);
    // lanes 0..2: one division + square root instruction stream for the three
    const double num = w.lane == 0 ? s_fr : (w.lane == 1 ? s_gr : s_td);
    const double sd = sqrt(num / (double)(w.lane == 2 ? w.M : w.K));
    const int pos = w.lane == 0 ? P::sd0 : (w.lane == 1 ? P::sd1 : P::sd2);
    if (w.lane < 3) w.scrL[pos] = sd;
}

// The rest of the observation once the tail has run: static entries, and for SO_SFJSP / MO_DFJSP the
// per-machine gap_ave statistics.  Leaves obs[0..n_obs) in the LDS scratch row.
template <int KC, int V, int RING = 2>
__device__ __forceinline__ void observe_finish(W<KC, V> &w) {
    const int M = w.M;
    if (V == FJSP_VARIANT_SO_SFJSP || V == kDyn) {
        // MO_DFJSP_breakdown.py:94-118: [DDT, M, S, ct_std, ratio_idle, cro_ave, cro_std, gap_ave, gap_std, gap_m_ave,
        //                                gap_m_std, dro_a, dro_e, drj_a, drj_e]
        // SO_SFJSP.py:64-83: [M_idle_ratio, ct_std, cro_ave, cro_std, ratio_idle, gap_ave, gap_std, gap_m_ave, gap_m_std]
        const uint32_t idle = ~w.busy & w.mmask;
        int nav = 0, nfav = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const bool has = w.fifo_cnt[c] > 0;
            nav += __builtin_popcountll(__ballot(has && (w.elig[c] & idle) != 0));
            nfav += __builtin_popcountll(__ballot(has && (w.fmask[c] & idle) != 0));
        }
        const double m_idle_ratio = (double)__builtin_popcount(idle) / (double)M;
        const double ratio_idle = (double)nfav / ((double)nav + 1e-08);
        // Machine.gap_ave of EVERY machine (class_FJSSP.py:144-146), three machines per pass: their gap rows
        // (ineligible entries +0.0, an exact identity of the running sum) go to the three LDS rows and lanes
        // 0..2 walk one row each.  In the step kernel the unprocessed matrix lives in HBM and lane 0 has just
        // updated one element of it, so that store is drained first.
        if (!w.single_job) wave_sync_global();
        double gave_m = 0.0;                       // lane m (< M): gap_ave of machine m
        for (int m0 = 0; m0 < M; m0 += 3) {
            const double v = gap_ave3<KC, V, RING>(w, m0, m0 + 1 < M ? m0 + 1 : -1, m0 + 2 < M ? m0 + 2 : -1);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (w.lane == m0 + q) gave_m = rld(v, q);
        }
        // gap_m_ave / gap_m_std over machines in ascending order (:78-80)
        if (w.lane < (uint32_t)w.KP) w.tdL[w.lane] = w.lane < M ? gave_m : 0.0;
        wave_sync();
        const int m8 = (M + 7) & ~7;
        const double gm_ave = lds_chain_sum<RING>(w.tdL, m8) / (double)M;
        wave_sync();
        { const double d = gave_m - gm_ave; if (w.lane < (uint32_t)w.KP) w.tdL[w.lane] = w.lane < M ? d * d : 0.0; }
        wave_sync();
        const double gm_std = sqrt(lds_chain_sum<RING>(w.tdL, m8) / (double)M);
        if (w.lane == 0) {
            if (V == kDyn) {
                w.scrL[0] = w.sstate[0]; w.scrL[1] = (double)M; w.scrL[2] = (double)w.n_orders; w.scrL[4] = ratio_idle;
                w.scrL[9] = gm_ave; w.scrL[10] = gm_std;
            } else {
                w.scrL[0] = m_idle_ratio; w.scrL[4] = ratio_idle; w.scrL[7] = gm_ave; w.scrL[8] = gm_std;
            }
        }
    } else if (is_so_v<V>) {
        if (w.lane == 0) w.scrL[0] = (double)M;
    }
    wave_sync();
}

// The whole observation by the environment's own wave (reset, fused rollout, arrival).  Returns tard_unproc.
template <int KC, int V, int RING = 2>
__device__ __forceinline__ long long observe(W<KC, V> &w, bool stats_only = false) {
    double frv[KC], grv[KC];
    const long long tard_unproc = observe_prepare<KC, V>(w, stats_only, frv, grv);
    if (stats_only) return tard_unproc;
    const uint32_t stride = 0;
    wave_sync();
    tail_pass<V, RING>(reinterpret_cast<unsigned char *>(w.scrL), stride, 1, w.KP);
    wave_sync();
    observe_deviations<KC, V>(w, frv, grv);
    wave_sync();
    observe_finish<KC, V, RING>(w);
    return V == FJSP_VARIANT_SO_SFJSP ? 0 : tard_unproc;      // SO_SFJSP never calls update_parameter: delay_time_sum_unprocessed stays 0
}

// state = [static, v(t), v(t) - v(t-1)]  (SO_FJSSP.py:71-72,257-258); updates obs_prev.
template <int KC, int V>
__device__ __forceinline__ void emit_state(W<KC, V> &w, double *state_out, bool zero_gap, float *x_lds = nullptr) {
    const int n_obs = w.n_obs, n_static = w.n_static;
    const double cur = w.lane < n_obs ? w.scrL[w.lane] : 0.0;
    const double gap = zero_gap ? cur - cur : cur - w.obs_prev_l;
    if (w.lane < n_obs) w.obs_prev_l = cur;
    if (x_lds) {                   // the state as the policy reads it: `.float()` of the f64 vector (MPPPO.py:274)
        if (w.lane < n_static) x_lds[w.lane] = (float)w.sstate[w.lane];
        if (w.lane < n_obs) { x_lds[n_static + w.lane] = (float)cur; x_lds[n_static + n_obs + w.lane] = (float)gap; }
    }
    if (state_out) {
        double *o = state_out + (size_t)w.env * w.state_size;
        if (w.lane < n_static) o[w.lane] = w.sstate[w.lane];
        if (w.lane < n_obs) { o[n_static + w.lane] = cur; o[n_static + n_obs + w.lane] = gap; }
    }
    wave_sync();
}

// SO_FJSSP.py:51-76 reset (fresh-object semantics; class_FJSSP.py:173-244 for one order).
// cached_obs: the observation of the reset state is a pure function of the instance (clock 0, nothing
// dispatched); reset_kernel leaves it in the instance record (i_obs0) and the autoreset path of step_kernel
// takes it from there instead of recomputing it.
template <int KC, int V>
__device__ __forceinline__ void init_episode(W<KC, V> &w, const DevBatch *b, double *state_out, bool cached_obs) {
    w.t = 0; w.step_count = 0; w.done = 0; w.status = 0;
    w.busy = 0; w.completion = 0; w.completion_last = 0;
    w.tard_done = 0; w.delay_sum = 0;
    w.tend_m = 0; w.mjob_m = -1;
    w.t_arr = 0; w.next_order = 1; w.pending = 0; w.obs_stale = 0;
    w.energy = 0; w.energy_last = 0; w.tlast_m = -1;
    if (is_mord_v<V>) {
        // per-environment copy of the reset-time fluid tables (they change at every later arrival)
        const Layout &L = b->L;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * kWave + w.lane;
            const int r = (int)((w.kB[c] >> 16) & 0xFFu);
            w.q0[c] = ((w.kB[c] >> 24) & 2u) ? (int)reinterpret_cast<const uint16_t *>(w.ir + L.i_ocnt)[r] : 0;
            // (a restart inside step_kernel arrives here with the PREVIOUS episode's last-arrival tables in the
            // wave state: always take the reset-time ones from the instance record)
            w.fmask[c] = reinterpret_cast<const uint32_t *>(w.ir + L.i_fmask)[k];
            w.rate_sum[c] = reinterpret_cast<const double *>(w.ir + L.i_rsum)[k];
            w.time_sum[c] = reinterpret_cast<const double *>(w.ir + L.i_tsum)[k];
            reinterpret_cast<uint32_t *>(w.er + L.e_fmask)[k] = w.fmask[c];
            reinterpret_cast<double *>(w.er + L.e_rsum)[k] = w.rate_sum[c];
            reinterpret_cast<double *>(w.er + L.e_tsum)[k] = w.time_sum[c];
            reinterpret_cast<uint32_t *>(w.er + L.e_q0)[k] = (uint32_t)w.q0[c];
            w.fifo_cnt[c] = 0;
        }
        const double *src = reinterpret_cast<const double *>(w.ir + L.i_col);
        double *dst = reinterpret_cast<double *>(w.er + L.e_col);
        for (int i = w.lane; i < w.K * w.MP * 2; i += kWave) dst[i] = src[i];
        for (int n = w.lane; n < w.njobs; n += kWave) w.jstL[n] = jst_pack(kNoSeq, kAbsent);
        w.seq_ctr = 0; w.n_unassigned = 0;
        wave_sync();
        wave_sync_global();
        order_arrive<KC, V>(w, b, 0);                                                         // class_FJSSP.py:225
    } else {
        w.n_unassigned = w.njobs;
        w.seq_ctr = (uint32_t)w.njobs;
        if (w.jreg) {
#pragma unroll
            for (int c = 0; c < KC; ++c) w.jsk[c] = jst_pack((w.kB[c] >> 16) & 0xFFu, 0u);                 // (job n = kind n)
        } else {
            for (int n = w.lane; n < w.njobs; n += kWave) w.jstL[n] = jst_pack((uint32_t)n, 0u);  // class_FJSSP.py:225
        }
    }
    if (w.single_job) {
        // :304 unprocessed = arrival: nothing assigned yet
#pragma unroll
        for (int c = 0; c < KC; ++c) { w.asg[c] = 0xFFu; reinterpret_cast<uint8_t *>(w.er + w.e_asg)[c * kWave + w.lane] = 0xFFu; }
        wave_sync();
    } else {
        for (int i = w.lane; i < w.K * w.MP; i += kWave) w.unp[i] = w.col_i[2 * i];          // :304
        wave_sync();
        wave_sync_global();   // the step kernel keeps the unprocessed matrix in HBM; later lanes gather from it
    }
    compute_params<KC, V>(w);
    if (cached_obs) {
        w.obs_prev_l = w.lane < w.n_obs ? reinterpret_cast<const double *>(w.ir + b->L.i_obs0)[w.lane] : 0.0;
        return;
    }
    observe<KC, V>(w);                      // delay_time_sum_unprocessed is 0-relevant only after a step
    if (w.env < b->n_inst && w.lane < w.n_obs)          // first environment of its instance: publish the reset observation
        reinterpret_cast<double *>(b->inst + (size_t)w.inst * b->L.i_stride + b->L.i_obs0)[w.lane] = w.scrL[w.lane];
    emit_state<KC, V>(w, state_out, true);
}

// First half of step() (SO_FJSSP.py:168-250): rule pair -> (operation type, machine), dispatch, event loop.
// compute_params() must be current on entry.  Returns true when the step goes on to its second half now
// (false: an error status was set, or -- multi-order batches -- the env parked at an order arrival).
template <int KC, int V, int RING = 2>
__device__ __forceinline__ bool env_step_decide(W<KC, V> &w, const DevBatch *b, int a0, int a1, int *k_out, int *m_out) {
    const bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES, is_sf = V == FJSP_VARIANT_SO_SFJSP;
    *k_out = -1; *m_out = -1;
    if (is_mo) {                     // flat action -> self.actions[action] (MO_FJSSP_discretes.py:26,92)
        if (a0 >= 18) { w.status |= FJSP_ST_BAD_TASK_RULE; return false; }   // IndexError
        a1 = a0 % 3; a0 = a0 / 3;
    } else if (is_sf) {              // SO_SFJSP.py:25,87-88
        if (a0 >= 20) { w.status |= FJSP_ST_BAD_TASK_RULE; return false; }
        a1 = a0 % 5; a0 = a0 / 5;
    }
    const uint32_t idle = ~w.busy & w.mmask;
    const int k_sel = task_select<KC, V>(w, a0, idle);
    STAMP(w, 2);
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 7
    w.rng_calls += (uint32_t)k_sel; *k_out = k_sel; return false;      // diagnostic: stop after task_select
#endif
    int pm = 0, en_sel = 0;
    double un_sel = 0.0;
    const int m_sel = k_sel >= 0 ? machine_select<KC, V, RING>(w, a1, k_sel, idle, &pm, &un_sel, &en_sel) : -1;
    STAMP(w, 3);
    *k_out = k_sel; *m_out = m_sel;
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 8
    w.rng_calls += (uint32_t)(k_sel + m_sel + pm) + (uint32_t)un_sel; return false;   // diagnostic: stop after machine_select
#endif
    if (k_sel < 0 || m_sel < 0) return false;       // status carries the MyError / undefined-behaviour bit
    dispatch_and_advance<KC, V>(w, b, k_sel, m_sel, pm, un_sel, en_sel);
    STAMP(w, 4);
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 9
    return false;                                                                 // diagnostic: stop after dispatch_and_advance
#endif
    return !(is_mord_v<V> && w.pending);            // an order arrived: the step is finished by arrival_kernel
}

// Reward and bookkeeping of step() (SO_FJSSP.py:259-265) once the observation is out.
template <int KC, int V>
__device__ __forceinline__ double step_reward(W<KC, V> &w, const double *mo, long long tard_unproc) {
    const bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES, is_sf = V == FJSP_VARIANT_SO_SFJSP;
    const long long delay_new = w.tard_done + tard_unproc;                   // :259
    const long long delta = delay_new - w.delay_sum;
    const int dc = w.completion_last - w.completion;
    w.delay_sum = delay_new;                                                 // :263
    w.completion_last = w.completion;
    if (V == kDyn) {                 // MO_DFJSP_breakdown.py:430-447 compute_reward(reward_policy, completion, tardiness, energy)
        const double de = (double)(w.energy_last - w.energy);
        w.energy_last = w.energy;
        const double pol = mo ? mo[0] : 1.0;
        if (pol == 0.0) return (double)dc;
        if (pol == 1.0) return (double)(-delta);
        if (pol == 2.0) return de;
        if (pol == 3.0) {
            const double cn = mo[1], tn = mo[2], en = mo[3];
            if (tn > 0.0) return (double)dc / cn + (double)(-delta) / tn + de / en;
            return (double)dc / cn + de / en;
        }
        w.status |= FJSP_ST_BAD_TASK_RULE;                                   // MyError :447
        return 0.0;
    }
    if (is_sf) return (double)dc / w.fluid_completed_time;                   // SO_SFJSP.py:216-220 (dc = -(C - C_last))
    if (!is_mo) return (double)(-delta);                                     // :328 (exact integer)
    // MO_FJSSP_discretes.py:232-244 compute_reward(weight_vector, completion, tardiness)
    const double w0 = mo ? mo[0] : 0.0, w1 = mo ? mo[1] : 1.0, cn = mo ? mo[2] : 0.0, tn = mo ? mo[3] : 0.0;
    if (cn > 0.0 && tn > 0.0) return (double)dc / cn * w0 + (double)(-delta) / tn * w1;
    if (w1 == 1.0) return (double)(-delta);
    if (w0 == 1.0) return (double)dc;
    w.status |= FJSP_ST_BAD_TASK_RULE;                                       // MyError :244
    return 0.0;
}

// Second half of step() (SO_FJSSP.py:252-265) by the environment's own wave: observation, reward, bookkeeping.
// A step that hands no state back skips the observation and marks the kept v(t-1) stale (obs_refresh).
template <int KC, int V, int RING = 2>
__device__ __forceinline__ double env_step_finish(W<KC, V> &w, const double *mo, double *state_out, bool need_obs = true,
                                                  float *x_lds = nullptr) {
    w.step_count++;                                                          // :252
    compute_params<KC, V>(w);
    STAMP(w, 5);
    const long long tard_unproc = observe<KC, V, RING>(w, !need_obs);         // :256
    STAMP(w, 6);
    if (need_obs) { emit_state<KC, V>(w, state_out, false, x_lds); w.obs_stale = 0; }
    else w.obs_stale = 1;
    STAMP(w, 7);
    return step_reward<KC, V>(w, mo, tard_unproc);
}

// v(t-1) of the state vector (SO_FJSSP.py:257-258) is the observation of the environment as the previous
// step left it, i.e. as this step finds it.  Steps that hand no state back do not keep it current; the first
// step that wants a state again rebuilds it here, before anything is dispatched.
template <int KC, int V>
__device__ __forceinline__ void obs_refresh(W<KC, V> &w) {
    observe<KC, V>(w);
    w.obs_prev_l = w.lane < w.n_obs ? w.scrL[w.lane] : 0.0;
    w.obs_stale = 0;
    wave_sync();
}

template <int KC, int V, int RING = 2>
__device__ __forceinline__ double env_step(W<KC, V> &w, const DevBatch *b, int a0, int a1, const double *mo,
                                           double *state_out, int *k_out, int *m_out, bool need_obs = true, float *x_lds = nullptr) {
    if (need_obs && w.obs_stale) obs_refresh<KC, V>(w);
    if (!env_step_decide<KC, V, RING>(w, b, a0, a1, k_out, m_out)) return 0.0;
    return env_step_finish<KC, V, RING>(w, mo, state_out, need_obs, x_lds);
}

// ------------------------------------------------------------------------ kernels
// update_fluid_parameter (class_FJSSP.py:282-306): one thread per (instance, k).
__global__ void fluid_tables_kernel(DevBatch b) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= b.n_inst * b.KP) return;
    const int inst = idx / b.KP, k = idx % b.KP;
    const InstHeader h = *inst_ptr<const InstHeader>(b, inst, 0);
    const int MP = b.MP;
    const uint16_t *p = inst_ptr<const uint16_t>(b, inst, b.L.i_p) + (size_t)k * MP;
    const double *x = inst_ptr<const double>(b, inst, b.L.i_x) + (size_t)k * MP;
    double *col = inst_ptr<double>(b, inst, b.L.i_col) + (size_t)k * MP * 2;
    uint32_t fm = 0;
    double s = 0.0;
    const bool valid = k < h.K;
    // Q0 of the reset-time LP: the jobs of the kind; with several orders only those of order 0
    double q0 = (double)(inst_ptr<const uint32_t>(b, inst, b.L.i_kA)[k] >> 16);
    if (b.mord && valid) {
        const int r = (int)((inst_ptr<const uint32_t>(b, inst, b.L.i_kB)[k] >> 16) & 0xFFu);
        q0 = (double)inst_ptr<const uint16_t>(b, inst, b.L.i_ocnt)[r];
    }
    for (int m = 0; m < h.M; ++m) {
        const int pm = p[m];
        double r = 0.0;
        if (valid && pm > 0) {
            const double xv = x[m];
            r = xv * (1.0 / (double)pm);                // :164, :288-289
            if (xv != 0) fm |= 1u << m;                 // :290-292
            s = s + r;                                  // :294 (ascending m)
        }
        col[2 * m + 1] = r;
    }
    inst_ptr<uint32_t>(b, inst, b.L.i_fmask)[k] = fm;
    inst_ptr<double>(b, inst, b.L.i_rsum)[k] = valid ? s : 0.0;
    inst_ptr<double>(b, inst, b.L.i_tsum)[k] = valid ? 1.0 / s : 0.0;       // :295
    for (int m = 0; m < h.M; ++m) {
        double a = 0.0;
        if (valid && p[m] > 0) a = (q0 * col[2 * m + 1]) / s;               // :300-302
        col[2 * m] = a;
        if (b.grp) inst_ptr<double2>(b, inst, b.L.i_colm)[m * 64 + k] = make_double2(a, col[2 * m + 1]);      // machine-major copy
    }
    if (b.grp) {
        // the packed copy the group kernels load (Layout::i_op); one job per kind: job index == kind index
        const uint32_t kb = inst_ptr<const uint32_t>(b, inst, b.L.i_kB)[k];
        const int due = valid ? inst_ptr<const int32_t>(b, inst, b.L.i_due)[(kb >> 16) & 0xFFu] : 0;
        unsigned char *slot = inst_ptr<unsigned char>(b, inst, b.L.i_op) + (size_t)(k >> 4) * 512;
        // (the fourth word belongs to the host packer: slot 0, lane m = operation types machine m can process)
        uint32_t *a = reinterpret_cast<uint32_t *>(slot) + 4 * (k & 15);
        a[0] = kb; a[1] = (inst_ptr<const uint32_t>(b, inst, b.L.i_elig)[k] & 0xFFu) | (fm << 8); a[2] = (uint32_t)due;
        inst_ptr<uint2>(b, inst, b.L.i_op8)[k] = make_uint2(a[0], a[1]);              // the same two words, 8 bytes apart (large-batch build)
        reinterpret_cast<double2 *>(slot + 256)[k & 15] = make_double2(valid ? s : 0.0, valid ? 1.0 / s : 0.0);
    }
}

template <int KC, int V>
__global__ __launch_bounds__(256) void reset_kernel(DevBatch b, const uint8_t *mask, double *state_out) {
    const int wave = uni((int)(threadIdx.x >> 6));   // wave-uniform: keeps every record pointer in SGPRs
    const int env = blockIdx.x * (blockDim.x >> 6) + wave;
    if (env >= b.N) return;
    if (mask && mask[env] == 0) return;
    W<KC, V> w;
    open_env<KC, V>(w, &b, env, fjsp_lds + wave * lds_bytes_per_wave(b.JP, b.MP, b.KP, false), false, false);
    // rng_calls survives a reset (the reference's global `random` state does too)
    w.rng_calls = env_ptr<const EnvScalars>(b, env, 0)->rng_calls;
    w.obs_prev_l = 0.0;
    init_episode<KC, V>(w, &b, state_out, false);
    store_dynamic<KC, V>(w, false);
}

// One step of every environment.  Single-order variants share the observation tail inside the workgroup
// (observe_tail above): every live wave of a workgroup passes the same four barriers, whatever happened to its
// environment (finished episode, invalid rule), so nothing below returns between the first barrier and the last.
// SJ: one job per kind in every instance of the batch (compile-time: the single-job kernel carries none of the list walks,
// statistics rows or their registers)
template <int KC, int V, bool SJ>
// (four chunks of per-lane operation state do not fit 128 VGPRs: K > 128 runs at half the occupancy instead of spilling)
__global__ __launch_bounds__(256, KC >= 4 ? 2 : (KC == 2 ? 3 : 4)) void step_kernel(DevBatch b, const uint8_t *actions, const double *mo, int autoreset,
                                                      double *state_out, double *reward_out, uint8_t *done_out,
                                                      int16_t *trace_km, uint8_t *ready) {
    constexpr bool SHARED = FJSP_SHARED_TAIL && !is_mord_v<V>;
    const int wave = uni((int)(threadIdx.x >> 6));   // wave-uniform: keeps every record pointer in SGPRs
    const int env_raw = blockIdx.x * (blockDim.x >> 6) + wave;
    // the waves past the last environment of a partial workgroup address the last environment until their loads are
    // out and leave then: no kernel argument has to arrive before the state loads can be issued
    const int env = min(env_raw, b.N - 1);
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 5
    return;                                     // diagnostic: launch overhead only
#endif
    W<KC, V> w;
    STAMP_BEGIN(w);
    // The action pair (wave-uniform, 2-byte aligned: checked by the host entry points) comes through the VECTOR memory
    // path and is only moved to scalar registers after the state loads are out: a scalar load of it here would be waited
    // for at once -- scalar loads return out of order, every wait on one drains them all -- one full memory round trip
    // before the first state load could be issued.
    const uint32_t araw = reinterpret_cast<const uint16_t *>(actions)[env];
    const uint32_t lds_stride = (uint32_t)lds_bytes_per_wave(b.JP, b.MP, kWave * KC, false);
    open_env<KC, V, SJ ? 1 : 0>(w, &b, env, fjsp_lds + wave * lds_stride, false, true, true);
    if (env_raw >= b.N) return;                       // (a finished wave no longer counts at the workgroup's barriers)
    const int a0 = uni((int)(araw & 0xFFu)), a1 = uni((int)(araw >> 8));
    STAMP(w, 0);
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 4
    store_dynamic<KC, V>(w, false);                // diagnostic: state in / state out only
    return;
#endif
    if (is_mord_v<V> && ready) {
        // asynchronous arrival service (fjsp_env_step_async): environments parked at an order arrival sit this launch
        // out (ready = 0), one that arrival_kernel has just finished in this same call keeps the outputs it was given
        if (w.pending == 2) { if (w.lane == 0) env_ptr<EnvScalars>(b, env, 0)->pending = 0; return; }
        if (w.pending == 1) { if (w.lane == 0) ready[env] = 0; return; }
    }
    const bool need_obs = state_out != nullptr;
    bool go = true;                  // this wave's environment takes a step in this launch
    if (w.done) {
        if (autoreset != 1) {        // 0: flag the misuse; 2: idle silently (non-fused rollout fallback)
            if (autoreset == 0) w.status |= FJSP_ST_STEP_AFTER_DONE;
            go = false;
        } else {
            init_episode<KC, V>(w, &b, nullptr, true);
        }
    } else if (w.single_job || !w.stats_ok) {
        compute_params<KC, V>(w);            // a handful of selects; batches with longer lists found the statistics in the record
    }
    STAMP(w, 1);
#if defined(FJSP_ABLATE) && FJSP_ABLATE == 3
    store_dynamic<KC, V>(w, false);                // diagnostic: + compute_params
    return;
#endif
    int k_sel = -1, m_sel = -1;
    double reward = 0.0;
    const double *mo_e = mo ? mo + (size_t)env * 4 : nullptr;
    if constexpr (!SHARED) {
        if (go) reward = env_step<KC, V, 8>(w, &b, a0, a1, mo_e, state_out, &k_sel, &m_sel, need_obs);
        if (go && w.pending) {
            // an order arrived inside this step: park the env for the host LP service (fjsp_env.hip), which
            // finishes the step with arrival_kernel; the outputs of this env are written there
            if (w.lane == 0) {
                int16_t *stash = reinterpret_cast<int16_t *>(w.er + b.L.e_lpq) + 2 * b.KP;
                stash[0] = (int16_t)k_sel; stash[1] = (int16_t)m_sel;
            }
            // take a slot of the service's staging area and leave the LP inputs there: the host fetches the
            // inputs of all parked envs with one copy
            uint32_t slot = 0;
            if (w.lane == 0) { slot = atomicAdd(b.pending_count, 1u); if (slot < (uint32_t)b.N) b.pending_count[1 + slot] = (uint32_t)env; }
            slot = min(uniu(slot), (uint32_t)b.N - 1u);     // (N slots: an env parks at most once per service; see service_arrivals)
            wave_sync_global();
            const uint32_t *src = reinterpret_cast<const uint32_t *>(w.er + b.L.e_lpq);     // u16[2][KP] as KP words
            uint32_t *dst = reinterpret_cast<uint32_t *>(b.lp_in + (size_t)slot * 2 * b.KP);
            for (int i = w.lane; i < b.KP; i += kWave) dst[i] = src[i];
            if (ready && w.lane == 0) ready[env] = 0;
            store_dynamic<KC, V>(w, false, false);        // (statistics are stale until arrival_kernel finishes the step)
            return;
        }
    } else {
        if (go && need_obs && w.obs_stale) obs_refresh<KC, V>(w);
        if (go) go = env_step_decide<KC, V, 8>(w, &b, a0, a1, &k_sel, &m_sel);
#if defined(FJSP_ABLATE) && (FJSP_ABLATE == 7 || FJSP_ABLATE == 8 || FJSP_ABLATE == 9)
        store_dynamic<KC, V>(w, false);
        return;
#endif
        double frv[KC], grv[KC];
        long long tard_unproc = 0;
        if (go) {
            w.step_count++;                                                  // SO_FJSSP.py:252
            compute_params<KC, V>(w);
            STAMP(w, 5);
#if !(defined(FJSP_ABLATE) && FJSP_ABLATE == 2)
            tard_unproc = observe_prepare<KC, V>(w, !need_obs, frv, grv);
#endif
            STAMP(w, 6);
        }
#if !(defined(FJSP_ABLATE) && FJSP_ABLATE == 2)
        if (need_obs) {              // (uniform over the grid: a kernel argument)
            // the walker runs the sequential tail of every environment of the workgroup; slots whose wave has
            // nothing to observe (finished episode, error) announce an empty row
            const int nslots = min(4, b.N - (int)blockIdx.x * 4);
            const int walker = min((int)(blockIdx.x & 3u), nslots - 1);
            if (!go && w.lane == 0) w.hdrL[H_N8] = 0;
            tail_sync<true>();
            STAMP(w, 7);
            if (wave == walker) tail_pass<V, 8>(fjsp_lds, lds_stride, nslots, kWave * KC);
            tail_sync<true>();
            STAMP(w, 8);
            if (go) { observe_deviations<KC, V>(w, frv, grv); wave_sync(); }     // (the standard deviations: no walk, no barrier)
            STAMP(w, 9);
            STAMP(w, 10);
            if (go) {
                observe_finish<KC, V, 8>(w);
#if !(defined(FJSP_ABLATE) && FJSP_ABLATE == 1)
                emit_state<KC, V>(w, state_out, false);
#endif
                STAMP(w, 11);
            }
        } else if (go) {
            w.obs_stale = 1;
        }
#endif
        if (go) reward = step_reward<KC, V>(w, mo_e, V == FJSP_VARIANT_SO_SFJSP ? 0 : tard_unproc);
    }
    if (w.lane == 0) {
        if (reward_out) reward_out[env] = reward;
        if (done_out) done_out[env] = (uint8_t)w.done;
        if (trace_km) { trace_km[(size_t)env * 2] = (int16_t)k_sel; trace_km[(size_t)env * 2 + 1] = (int16_t)m_sel; }
        if (ready) ready[env] = 1;
    }
    store_dynamic<KC, V>(w, false);
    STAMP(w, 12);
    STAMP_FLUSH(w);
}

// T fused steps per launch: the environment lives in registers + LDS for the whole episode.
template <int KC, int V>
__global__ __launch_bounds__(256, (KC == 1 && V == FJSP_VARIANT_SO_FJSSP) ? 4 : 1) void rollout_kernel(DevBatch b, const uint8_t *actions, const double *mo, int T,
                                                      int16_t *trace_km, double *reward_out, double *state_last) {
    const int wave = uni((int)(threadIdx.x >> 6));   // wave-uniform: keeps every record pointer in SGPRs
    const int env = blockIdx.x * (blockDim.x >> 6) + wave;
    if (env >= b.N) return;
    W<KC, V> w;
    open_env<KC, V>(w, &b, env, fjsp_lds + wave * lds_bytes_per_wave(b.JP, b.MP, b.KP, true), true, true);
    compute_params<KC, V>(w);
    for (int s = 0; s < T; ++s) {
        const size_t o = (size_t)s * b.N + env;
        int k_sel = -1, m_sel = -1;
        double reward = 0.0;
        const bool live = !w.done && !(w.status & (FJSP_ST_BAD_TASK_RULE | FJSP_ST_BAD_MACHINE_RULE | FJSP_ST_NO_EVENT));
        if (live) {
            const int a0 = actions[o * 2], a1 = actions[o * 2 + 1];
            reward = env_step<KC, V>(w, &b, uni(a0), uni(a1), mo ? mo + (size_t)env * 4 : nullptr, state_last, &k_sel, &m_sel,
                                     state_last != nullptr);
        }
        if (w.lane == 0) {
            if (trace_km) { trace_km[o * 2] = (int16_t)k_sel; trace_km[o * 2 + 1] = (int16_t)m_sel; }
            if (reward_out) reward_out[o] = reward;
        }
    }
    store_dynamic<KC, V>(w, true);
}

// ------------------------------------------------------------------ policy inside the launch
// Actor forward for a batch of states, one wavefront per state, 16 per workgroup, weights in LDS: the per-step
// counterpart of what rollout_policy_kernel evaluates in place (same device function, same arithmetic).
__global__ __launch_bounds__(1024) void actor_forward_kernel(ActorParams ap, const double *state, int n, float *probs) {
    float *lds = reinterpret_cast<float *>(fjsp_lds);
    actor_lds_fill(lds, ap, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();
    const int wave = uni((int)(threadIdx.x >> 6)), lane = (int)__lane_id();
    const int row = blockIdx.x * 16 + wave;
    if (row >= n) return;
    float *xs = lds + actor_lds_floats(ap.S) + (size_t)wave * (32 + kActorH + kActorAP);
    float *hs = xs + 32, *ps = hs + kActorH;
    if (lane < ap.S) xs[lane] = (float)state[(size_t)row * ap.S + lane];
    actor_wave_sync();
    actor_probs(lds, xs, hs, ps, ap.S, ap.A);
    if (lane < ap.A) probs[(size_t)row * ap.A + lane] = ps[lane];
}

// The T-step rollout with the actor inside the launch (replaces the per-step loop MPPPO.py:245-252: policy
// inference, sampling, env.step, buffer append): one wavefront per environment for the whole episode, sixteen
// per workgroup sharing the actor's weights in LDS.  Per step: actor_probs on the current state, sample_action
// by lane 0 (the counter-based stream of fjsp_policy_sample: same seed, same actions as the per-step path),
// the environment step, the buffer row.  Finished environments idle; their rows are marked invalid.
template <int KC, int V>
__global__ __launch_bounds__(1024) void rollout_policy_kernel(DevBatch b, ActorParams ap, PolicyRolloutIO io, const double *mo, int T) {
    float *lds = reinterpret_cast<float *>(fjsp_lds);
    actor_lds_fill(lds, ap, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();                                   // (the only workgroup barrier: waves may leave after it)
    const int wave = uni((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * 16 + wave;
    if (env >= b.N) return;
    const uint32_t env_stride = (uint32_t)lds_bytes_per_wave(b.JP, b.MP, b.KP, false);
    unsigned char *wave_lds = fjsp_lds + ((actor_lds_floats(ap.S) * 4 + 255) & ~(size_t)255) +
                              (size_t)wave * (env_stride + (32 + kActorH + kActorAP) * 4);
    float *xs = reinterpret_cast<float *>(wave_lds + env_stride);
    float *hs = xs + 32, *ps = hs + kActorH;
    W<KC, V> w;
    open_env<KC, V>(w, &b, env, wave_lds, false, true);
    compute_params<KC, V>(w);
    const int S = ap.S, A = ap.A, N = b.N;
    if (w.lane < S) xs[w.lane] = (float)io.state_in[(size_t)env * S + w.lane];
    const float eps = io.epsilon[0];
    const uint64_t seed = io.seed[0];
    const double *mo_e = mo ? mo + (size_t)env * 4 : nullptr;
    wave_sync();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * N + env;
        const bool live = !w.done && !(w.status & (FJSP_ST_BAD_TASK_RULE | FJSP_ST_BAD_MACHINE_RULE | FJSP_ST_NO_EVENT));
        if (!live) {
            if (w.done) w.status |= FJSP_ST_STEP_AFTER_DONE;       // what the per-step loop flags for the same launches
            // (rows of finished environments are masked by `valid`; they still get finite contents -- the last state,
            // like the per-step loop leaves there -- because masked arithmetic multiplies them by zero)
            if (w.lane < S) { io.o_state[row * S + w.lane] = xs[w.lane]; io.o_next[row * S + w.lane] = xs[w.lane]; }
            if (w.lane == 0) {
                io.o_valid[row] = 0.0f; io.o_reward[row] = 0.0f; io.o_done[row] = 1.0f;
                io.o_actions[row * 2] = 0.0f; io.o_actions[row * 2 + 1] = 0.0f; io.o_flat[row] = 0.0f; io.o_logp[row] = 0.0f;
            }
            continue;
        }
        if (w.lane < S) io.o_state[row * S + w.lane] = xs[w.lane];
        actor_probs(lds, xs, hs, ps, S, A);
        int action = 0;
        float logp = 0.0f;
        if (w.lane == 0) {
            const SampledAction sa = sample_action(ps, A, eps, seed, (uint64_t)t, env);
            action = sa.action; logp = sa.log_prob;
        }
        action = uni(action);
        const int a0 = io.pair_div > 0 ? action / io.pair_div : action, a1 = io.pair_div > 0 ? action % io.pair_div : 0;
        int k_sel, m_sel;
        const double reward = env_step<KC, V>(w, &b, a0, a1, mo_e, io.state_last, &k_sel, &m_sel, true, xs);
        if (w.lane < S) io.o_next[row * S + w.lane] = xs[w.lane];
        if (w.lane == 0) {
            io.o_actions[row * 2] = (float)a0; io.o_actions[row * 2 + 1] = (float)a1;
            io.o_reward[row] = (float)reward; io.o_done[row] = (float)w.done; io.o_valid[row] = 1.0f;
            io.o_flat[row] = (float)action; io.o_logp[row] = logp;
        }
    }
    store_dynamic<KC, V>(w, false, false);
}

// Multi-order: finish the step of every env parked at an order arrival.  The host service has solved the
// fluid LP of the env's live state (class_FJSSP.py:239) and left x in the env record; this kernel runs
// update_fluid_parameter (:282-306) for that env, then the second half of step().
template <int KC, int V>
__global__ __launch_bounds__(256, KC >= 4 ? 2 : 4) void arrival_kernel(DevBatch b, const double *mo, int n_pending, const uint32_t *ids, const double *x_list,
                                                         double *state_out, double *reward_out, uint8_t *done_out, int16_t *trace_km,
                                                         uint8_t *ready, int mark_resumed, const uint32_t *n_dev) {
    const int wave = uni((int)(threadIdx.x >> 6));
    const int idx = blockIdx.x * (blockDim.x >> 6) + wave;
    // (device LP service: the number of parked environments never visits the host; the grid covers the whole batch)
    if (n_dev) n_pending = (int)min(*n_dev, (uint32_t)b.N);
    if (idx >= n_pending) return;
    const int env = (int)ids[idx];
    W<KC, V> w;
    open_env<KC, V>(w, &b, env, fjsp_lds + wave * lds_bytes_per_wave(b.JP, b.MP, b.KP, false), false, true);
    const Layout &L = b.L;
    const double *xin = x_list + (size_t)idx * b.KP * b.MP;
    const uint16_t *lpq = reinterpret_cast<const uint16_t *>(w.er + L.e_lpq);
    double *col = reinterpret_cast<double *>(w.er + L.e_col);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = c * kWave + w.lane;
        const bool valid = k < w.K;
        const double q0 = valid ? (double)lpq[k] : 0.0;
        uint32_t fm = 0;
        double ssum = 0.0;
        for (int m = 0; m < w.M; ++m) {
            const int o = k * w.MP + m;
            const int pm = valid ? w.p_i[o] : 0;
            double r = 0.0;
            if (pm > 0) {
                const double xv = xin[o];
                r = xv * (1.0 / (double)pm);                // :164, :288-289
                if (xv != 0) fm |= 1u << m;                 // :290-292
                ssum = ssum + r;                            // :294 (ascending m)
            }
            if (valid) col[2 * o + 1] = r;
        }
        for (int m = 0; m < w.M; ++m) {
            const int o = k * w.MP + m;
            if (!valid) continue;
            double a = 0.0;
            if (w.p_i[o] > 0) a = (q0 * col[2 * o + 1]) / ssum;              // :300-302
            col[2 * o] = a;
            w.unp[o] = a;                                                    // :304 unprocessed restarts at arrival
        }
        w.fmask[c] = fm; w.rate_sum[c] = valid ? ssum : 0.0; w.time_sum[c] = valid ? 1.0 / ssum : 0.0;   // :295
        w.q0[c] = valid ? (int)lpq[k] : 0;
        reinterpret_cast<uint32_t *>(w.er + L.e_fmask)[k] = fm;
        reinterpret_cast<double *>(w.er + L.e_rsum)[k] = w.rate_sum[c];
        reinterpret_cast<double *>(w.er + L.e_tsum)[k] = w.time_sum[c];
        reinterpret_cast<uint32_t *>(w.er + L.e_q0)[k] = (uint32_t)w.q0[c];
    }
    wave_sync_global();
    w.pending = mark_resumed ? 2 : 0;       // (asynchronous service: the step launch of this same call must leave this env alone)
    const double reward = env_step_finish<KC, V>(w, mo ? mo + (size_t)env * 4 : nullptr, state_out);
    if (w.lane == 0) {
        const int16_t *stash = reinterpret_cast<const int16_t *>(w.er + L.e_lpq) + 2 * b.KP;
        if (reward_out) reward_out[env] = reward;
        if (done_out) done_out[env] = (uint8_t)w.done;
        if (trace_km) { trace_km[(size_t)env * 2] = stash[0]; trace_km[(size_t)env * 2 + 1] = stash[1]; }
        if (ready) ready[env] = 1;
    }
    store_dynamic<KC, V>(w, false);
}

// attribute read-back: one thread per environment
__global__ void read_kernel(DevBatch b, int64_t *delay, int32_t *makespan, int32_t *completion, int32_t *step_time,
                            int32_t *step_count, uint8_t *done, uint32_t *status) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= b.N) return;
    const EnvScalars s = *env_ptr<const EnvScalars>(b, env, 0);
    const int M = inst_ptr<const InstHeader>(b, env % b.n_inst, 0)->M;
    const int32_t *tend = env_ptr<const int32_t>(b, env, b.L.e_tend);
    int mx = 0;
    for (int m = 0; m < M; ++m) mx = max(mx, tend[m]);                       // SO_FJSSP.py:427 max time_end
    if (delay) delay[env] = s.delay_sum;
    if (makespan) makespan[env] = mx;
    if (completion) completion[env] = s.completion;
    if (step_time) step_time[env] = s.t;
    if (step_count) step_count[env] = s.step_count;
    if (done) done[env] = (uint8_t)s.done;
    if (status) status[env] = s.status;
}

// ------------------------------------------------------------------ host launchers
static inline dim3 grid_for(int N) { return dim3((unsigned)((N + 3) / 4)); }
// Dynamic LDS beyond the 64 KB default must be allowed per kernel (up to the 160 KB of a gfx950 CU; create refuses
// batches beyond that, step_lds_bytes()).
template <class K>
static inline void allow_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
size_t step_lds_bytes(const DevBatch &b) { return 4 * lds_bytes_per_wave(b.JP, b.MP, b.KP, false); }

int launch_fluid_tables(const DevBatch &b, hipStream_t st) {
    const int n = b.n_inst * b.KP;
    hipLaunchKernelGGL(fluid_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// one instantiation per (chunk count, environment variant): the variant is a compile-time
// parameter so SO_FJSSP's kernels carry none of the subclasses' code or registers
template <int V, class F>
static int dispatch_kc(int kc, F &&f) {
    switch (kc) {
    case 1: f(std::integral_constant<int, 1>{}, std::integral_constant<int, V>{}); return 0;
    case 2: f(std::integral_constant<int, 2>{}, std::integral_constant<int, V>{}); return 0;
    case 4: f(std::integral_constant<int, 4>{}, std::integral_constant<int, V>{}); return 0;
    default: return -1;
    }
}
template <class F>
static int dispatch(const DevBatch &b, F &&f) {
    if (b.variant == FJSP_VARIANT_MO_DFJSP) return dispatch_kc<kDyn>(b.KC, f);
    if (b.mord) return dispatch_kc<kMord>(b.KC, f);
    switch (b.variant) {
    case FJSP_VARIANT_SO_FJSSP: return dispatch_kc<FJSP_VARIANT_SO_FJSSP>(b.KC, f);
    case FJSP_VARIANT_SO_SFJSP: return dispatch_kc<FJSP_VARIANT_SO_SFJSP>(b.KC, f);
    case FJSP_VARIANT_MO_FJSSP_DISCRETES: return dispatch_kc<FJSP_VARIANT_MO_FJSSP_DISCRETES>(b.KC, f);
    default: return -1;
    }
}

int launch_reset(const DevBatch &b, const uint8_t *mask, double *state, hipStream_t st) {
    const size_t lds = step_lds_bytes(b);
    if (dispatch(b, [&](auto kc, auto v) {
            allow_lds(&reset_kernel<decltype(kc)::value, decltype(v)::value>, lds);
            hipLaunchKernelGGL((reset_kernel<decltype(kc)::value, decltype(v)::value>), grid_for(b.N), dim3(256), lds, st, b,
                               mask, state);
        }) != 0) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_step(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                uint8_t *done, int16_t *trace_km, hipStream_t st, uint8_t *ready) {
    if (b.grp && !ready) return launch_step_group(b, actions, mo, autoreset, state, reward, done, trace_km, st);
    const size_t lds = step_lds_bytes(b);
    if (dispatch(b, [&](auto kc, auto v) {
            constexpr int KC = decltype(kc)::value, V = decltype(v)::value;
            if constexpr (!is_mord_v<V>) {
                if (b.single_job) {
                    allow_lds(&step_kernel<KC, V, true>, lds);
                    hipLaunchKernelGGL((step_kernel<KC, V, true>), grid_for(b.N), dim3(256), lds, st, b, actions, mo, autoreset, state, reward,
                                       done, trace_km, ready);
                    return;
                }
            }
            allow_lds(&step_kernel<KC, V, false>, lds);
            hipLaunchKernelGGL((step_kernel<KC, V, false>), grid_for(b.N), dim3(256), lds, st, b, actions, mo, autoreset, state, reward, done,
                               trace_km, ready);
        }) != 0) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
size_t rollout_lds_bytes(const DevBatch &b) { return 4 * lds_bytes_per_wave(b.JP, b.MP, b.KP, true); }
int launch_rollout(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                   double *state_last, hipStream_t st) {
    if (b.grp) return launch_rollout_group(b, actions, mo, T, trace_km, reward, state_last, st);
    const size_t lds = rollout_lds_bytes(b);
    if (dispatch(b, [&](auto kc, auto v) {
            constexpr int KC = decltype(kc)::value, V = decltype(v)::value;
            allow_lds(&rollout_kernel<KC, V>, lds);
            hipLaunchKernelGGL((rollout_kernel<KC, V>), grid_for(b.N), dim3(256), lds, st, b, actions, mo, T, trace_km, reward,
                               state_last);
        }) != 0) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

size_t policy_rollout_lds_bytes(const DevBatch &b, int S) {
    return ((actor_lds_floats(S) * 4 + 255) & ~(size_t)255) +
           16 * (lds_bytes_per_wave(b.JP, b.MP, b.KP, false) + (32 + kActorH + kActorAP) * 4);
}
int launch_actor_forward(const ActorParams &ap, const double *state, int n, float *probs, hipStream_t st) {
    const size_t lds = actor_lds_floats(ap.S) * 4 + 16 * (32 + kActorH + kActorAP) * 4;
    allow_lds(&actor_forward_kernel, lds);
    hipLaunchKernelGGL(actor_forward_kernel, dim3((unsigned)((n + 15) / 16)), dim3(1024), lds, st, ap, state, n, probs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_rollout_policy(const DevBatch &b, const ActorParams &ap, const PolicyRolloutIO &io, const double *mo, int T, hipStream_t st) {
    if (b.mord || b.KC != 1) return -1;                 // order arrivals need the host LP service between steps; K <= 64
    const size_t lds = policy_rollout_lds_bytes(b, ap.S);
    if (dispatch(b, [&](auto kc, auto v) {
            constexpr int KC = decltype(kc)::value, V = decltype(v)::value;
            if constexpr (!is_mord_v<V> && KC == 1) {
                allow_lds(&rollout_policy_kernel<KC, V>, lds);
                hipLaunchKernelGGL((rollout_policy_kernel<KC, V>), dim3((unsigned)((b.N + 15) / 16)), dim3(1024), lds, st, b, ap, io, mo, T);
            }
        }) != 0) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_arrival(const DevBatch &b, const double *mo, int n_pending, const uint32_t *ids, const double *x_list, double *state,
                   double *reward, uint8_t *done, int16_t *trace_km, hipStream_t st, uint8_t *ready, bool mark_resumed, const uint32_t *n_dev) {
    const size_t lds = step_lds_bytes(b);
    const dim3 grid((unsigned)(((n_dev ? b.N : n_pending) + 3) / 4));
    auto go = [&](auto kc, auto v) {
        allow_lds(&arrival_kernel<decltype(kc)::value, decltype(v)::value>, lds);
        hipLaunchKernelGGL((arrival_kernel<decltype(kc)::value, decltype(v)::value>), grid, dim3(256), lds, st, b, mo,
                           n_pending, ids, x_list, state, reward, done, trace_km, ready, mark_resumed ? 1 : 0, n_dev);
    };
    const int rc = b.variant == FJSP_VARIANT_MO_DFJSP ? dispatch_kc<kDyn>(b.KC, go) : dispatch_kc<kMord>(b.KC, go);
    if (rc != 0) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_read(const DevBatch &b, int64_t *delay, int32_t *makespan, int32_t *completion, int32_t *step_time,
                int32_t *step_count, uint8_t *done, uint32_t *status, hipStream_t st) {
    hipLaunchKernelGGL(read_kernel, dim3((unsigned)((b.N + 255) / 256)), dim3(256), 0, st, b, delay, makespan,
                       completion, step_time, step_count, done, status);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#ifdef FJSP_STAMPS
extern "C" int fjsp_debug_read_stamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fjsp_stamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fjsp_stamp_acc), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace fjsp
