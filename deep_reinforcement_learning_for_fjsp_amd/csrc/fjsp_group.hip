// gfx950 (MI355X / CDNA4) kernels of the batched rule-dispatch FJSP environment, second family:
// ONE 16-LANE DPP ROW PER ENVIRONMENT, four environments per wavefront.
//
// fjsp_kernels.hip gives every environment a whole wavefront; on the 10x5 / Brandimarte workloads (one job per kind,
// K <= 64 operation types, M <= 8 machines) 40 of its 64 lanes carry an operation type, 5 a machine, and the
// wave-uniform part of a step (the scalar stream, the reductions, the serial sums) is paid per environment.  Here
//   * lane l of row g owns the operation types k = 16 s + l of environment g, s = 0..3 ("slots": registers);
//   * lanes 0..M-1 of the row double as its machine lanes, lanes 0..njobs-1 as its job lanes (state word of job l),
//     lane (8 + i) & 15 keeps observation i;
//   * what was wave-uniform becomes row-uniform and lives in VGPRs: a value is read from a row's lane n with one DPP
//     move (row_newbcast:n), from a computed lane with ds_bpermute_b32; reductions are four DPP steps inside the row
//     (quad_perm, quad_perm, row_half_mirror, row_mirror) and leave the result in every lane of the row;
//   * control flow is wave-uniform only ("does any row of this wave need ...", a ballot): inside, lanes are predicated;
//   * the strictly sequential f64 sums of the observation and of Machine.gap_ave are walked out of LDS rows by one
//     lane per sum, all rows of the wave side by side (the chains of four environments cost one instruction stream);
//   * the static per-operation data come from a packed copy of the instance rows (Layout::i_op: two 16-byte loads
//     per slot), the assigned-machine bytes of a lane's four slots share one word (asg_pos()).
// The records in HBM are the ones of fjsp_kernels.hip: reset, the fused policy rollout, read-back and every other
// entry point keep working on the same batch, and a batch can be stepped by either family (FJSP_STEP_IMPL=wave).
//
// Reference restated (paths relative to the reference root), single-job form (every per-(r, j) list has at most one
// member, job index == kind index):
//   environments/SO_FJSSP.py:99-166   update_parameter -> g_params
//   environments/SO_FJSSP.py:168-250  step, first half -> g_task_select, g_machine_select, g_dispatch_advance
//   environments/SO_FJSSP.py:267-322  task_select, machine_select (class_FJSSP.py:137-146 gap, gap_ave)
//   environments/SO_FJSSP.py:78-97, 252-265 state_extract, reward -> g_observe, g_reward
//   environments/MO_FJSSP_discretes.py:26,88-244 flat action, three machine rules, weighted reward
// Compiled with -ffp-contract=off: a*b+c must round twice like CPython.
#include <hip/hip_runtime.h>

#include "../../include/fjsp_amd.h"
#include "fjsp_common.h"
#include "fjsp_device.h"

#pragma clang fp contract(off)

namespace fjsp {
namespace grp {

constexpr int GS = 4;       // operation-type slots per lane
constexpr int kNone = 255;  // "no lane" in a first-hit reduction

#define GDEV __device__ __forceinline__

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// ---- row primitives
template <int N>
GDEV int bc(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, false); }       // row_newbcast:N
template <int N>
GDEV uint32_t bcu(uint32_t v) { return (uint32_t)bc<N>((int)v); }
template <int N>
GDEV long long bcl(long long v) { return __builtin_amdgcn_update_dpp(0ll, v, 0x150 + N, 0xF, 0xF, false); }
template <int N>
GDEV double bcd(double v) { return __longlong_as_double(bcl<N>(__double_as_longlong(v))); }
// the value lane `idx` (0..15, any per-lane value) of the caller's row holds
GDEV int gread(int v, int idx, int gb) { return __builtin_amdgcn_ds_bpermute((gb + idx) << 2, v); }
GDEV uint32_t greadu(uint32_t v, int idx, int gb) { return (uint32_t)gread((int)v, idx, gb); }
GDEV int gsum(int v) {
    v += DPP(v, 0xB1, 0); v += DPP(v, 0x4E, 0); v += DPP(v, 0x141, 0); v += DPP(v, 0x140, 0);
    return v;
}
GDEV int gmin(int v) {
    v = min(v, DPP(v, 0xB1, 0x7fffffff)); v = min(v, DPP(v, 0x4E, 0x7fffffff));
    v = min(v, DPP(v, 0x141, 0x7fffffff)); v = min(v, DPP(v, 0x140, 0x7fffffff));
    return v;
}
GDEV uint32_t gmaxu(uint32_t v) {
    v = max(v, (uint32_t)DPP((int)v, 0xB1, 0)); v = max(v, (uint32_t)DPP((int)v, 0x4E, 0));
    v = max(v, (uint32_t)DPP((int)v, 0x141, 0)); v = max(v, (uint32_t)DPP((int)v, 0x140, 0));
    return v;
}
// the row's 16 bits of a wave ballot
GDEV uint32_t gballot(bool p, int gb) { return (uint32_t)(__ballot(p) >> gb) & 0xFFFFu; }
GDEV bool wave_any(bool p) { return __ballot(p) != 0ull; }
// LDS hand-off between lanes of one wave (LDS operations of a wave execute in order; only the compiler must be kept
// from moving accesses across the point; restricted to the LDS address space so global traffic is not drained)
GDEV void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// order-preserving 64-bit image of an f64 key (-0.0 folded onto +0.0: they compare equal in Python); keys are finite
GDEV uint64_t sortable(double z0) {
    const double z = z0 + 0.0;
    const uint32_t zh = (uint32_t)__double2hiint(z), zl = (uint32_t)__double2loint(z);
    const bool neg = (zh >> 31) != 0;
    return ((uint64_t)(neg ? ~zh : (zh | 0x80000000u)) << 32) | (uint64_t)(neg ? ~zl : zl);
}
GDEV uint64_t sortable_max_i32(int v) { return (uint64_t)((uint32_t)v ^ 0x80000000u) << 32; }
GDEV uint64_t sortable_min_i32(int v) { return (uint64_t)(~((uint32_t)v ^ 0x80000000u)) << 32; }
// row maximum of a 64-bit unsigned key: high words, then the low words among the lanes that tie on the high word
GDEV uint64_t gmax64(uint64_t v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mhi = gmaxu(hi);
    const uint32_t mlo = gmaxu(hi == mhi ? lo : 0u);
    return ((uint64_t)mhi << 32) | mlo;
}

// per-row LDS: RW rows of KS doubles (operands of the serial sums); KS = n8(kmax) + 2 keeps rows 16-byte aligned and the
// rows that are walked side by side in different banks ((2 KS) mod 64 is an odd multiple of 4 for every n8)
GDEV int row_stride(int kmax) { return ((kmax + 7) & ~7) + 2; }
__host__ __device__ inline size_t group_lds_bytes(int kmax, int MP) {
    const int ks = ((kmax + 7) & ~7) + 2, rw = MP < 2 ? 2 : MP;
    return (size_t)4 * rw * ks * 8 + 256;           // + the ring's read-ahead past the last row
}

// ------------------------------------------------------------------ row state
template <int V>
struct GE {
    int l, gb, env;
    bool live;                 // the row has an environment (the last wave of a batch may not be full)
    const unsigned char *ir;
    unsigned char *er;
    int MP;
    int nslots;                // wave-uniform: slots any row of this wave uses
    int n8w;                   // wave-uniform: operation types of the wave's largest instance, rounded up to 8 (length of a walk)
    // lane = operation types 16 s + l
    uint32_t kB[GS], em[GS];   // stage | J_r << 8 | kind << 16 | flags << 24;  elig | fmask << 8
    int due[GS];
    double rsum[GS], tsum[GS];
    uint32_t jw[GS];           // state word of the job of this slot's kind (copy of the job lane's)
    uint32_t asgw;             // byte s: machine operation type 16 s + l was assigned to (0xFF: not yet)
    // lane = job / machine / observation
    uint32_t jwl;
    int tend, mjob, mcnt;      // mcnt: operation types machine l can process (static)
    double obs_prev;           // lane (8 + i) & 15: observation i of the previous step
    // row-uniform
    int K, M, njobs;
    uint32_t mmask;
    int t, step_count, done, n_un, completion, completion_last;
    uint32_t status, seq_ctr, rng_calls, busy, misc;     // misc: EnvScalars t_arr's neighbour word (next_order | pending | obs_stale)
    int t_arr;
    long long tard_done, delay_sum;
    uint64_t env_seed;
    double *rows;              // this row's LDS rows
    int ks;
};

// per-slot results of update_parameter at the current clock (one job per kind: lists of at most one)
struct GP {
    bool in[GS], wait[GS], late[GS], cnte[GS];
    double de[GS];             // estimated delay of the kind's job at this stage (:136-139) = max_e = sum_e = urgency
    int da[GS];                // its actual delay t - due (:138) = max_a
};

template <int V>
GDEV void g_refresh_jw(GE<V> &e) {
#pragma unroll
    for (int s = 0; s < GS; ++s)
        if (s < e.nslots) e.jw[s] = greadu(e.jwl, (int)((e.kB[s] >> 16) & 0xFu), e.gb);
}

// SO_FJSSP.py:126-154 for lists of one (compute_params of fjsp_kernels.hip, single-job branch)
template <int V>
GDEV void g_params(const GE<V> &e, GP &p) {
    const double td = (double)e.t;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (s < e.nslots) {
            const uint32_t kb = e.kB[s], js = e.jw[s];
            const int j = (int)(kb & 0xFFu), nj = (int)(js & 0xFFu), d = e.due[s];
            const bool valid = ((kb >> 24) & 2u) != 0;
            const bool in = valid && nj <= j;                         // the job's stage-j task is still unassigned
            const double est = td + e.tsum[s];                        // :136,139 with task_index 0
            p.de[s] = est - (double)d;
            p.da[s] = e.t - d;                                        // :138
            p.in[s] = in;
            p.late[s] = in && e.t > d;                                // :134-135
            p.cnte[s] = in && est > (double)d;                        // :136-137
            p.wait[s] = in && nj == j && (js >> 8) != kNoSeq;         // head of job_now_list[(r, j)]
        } else {
            p.de[s] = 0.0; p.da[s] = 0; p.in[s] = false; p.late[s] = false; p.cnte[s] = false; p.wait[s] = false;
        }
    }
}

// random.choice replacement (fjsp_oracle.h): index into a list of length n; `take` rows consume one draw
template <int V>
GDEV int g_rng(GE<V> &e, bool take, int n) {
    const uint64_t u = splitmix64(e.env_seed + (uint64_t)e.rng_calls);
    if (take) e.rng_calls++;
    return (int)(((u >> 32) * (uint64_t)(uint32_t)n) >> 32);
}

// SO_FJSSP.py:267-298 task_select for every row with go; returns k or -1 (status set)
template <int V>
GDEV int g_task_select(GE<V> &e, const GP &p, bool go, int a0, uint32_t idle) {
    bool av[GS], fav[GS], C[GS];
    bool anyav_l = false, anyfav_l = false;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        av[s] = p.wait[s] && (e.em[s] & idle) != 0;
        fav[s] = p.wait[s] && ((e.em[s] >> 8) & idle) != 0;
        anyav_l = anyav_l || av[s]; anyfav_l = anyfav_l || fav[s];
    }
    const bool anyav = gballot(anyav_l, e.gb) != 0, anyfav = gballot(anyfav_l, e.gb) != 0;
    const bool noev = go && !anyav;
    if (noev) e.status |= FJSP_ST_NO_EVENT;
    const bool bad = go && anyav && a0 >= 6;
    if (bad) e.status |= FJSP_ST_BAD_TASK_RULE;                           // MyError :297
    const bool run = go && anyav && a0 < 6;
    // rules 1 and 2 prefer the types with an estimated / actual delay when any is available (:269-278)
    bool pre_l = false;
#pragma unroll
    for (int s = 0; s < GS; ++s) pre_l = pre_l || (av[s] && (a0 == 0 ? p.cnte[s] : p.late[s]));
    const bool anypre = gballot(pre_l && a0 <= 1, e.gb) != 0;
    const double dt = (double)e.t;                                        // gap_time (:237), one order: order_arrive_time = 0
    uint64_t key[GS], best = 0;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        bool c;
        uint64_t kv;
        if (a0 == 0) { c = anypre ? (av[s] && p.cnte[s]) : av[s]; kv = sortable(p.de[s]); }                 // :269-273
        else if (a0 == 1) { c = anypre ? (av[s] && p.late[s]) : av[s]; kv = anypre ? sortable_max_i32(p.da[s]) : sortable(p.de[s]); }
        else if (a0 == 2) {                                                                                   // :279-283
            c = anyfav ? fav[s] : av[s];
            const double fq = 1.0 - e.rsum[s] * dt;                       // fluid_unprocessed_number (:239-240), Q0 = 1 job
            kv = sortable((p.in[s] ? 1.0 : 0.0) - fq);                    // Tasks.gap class_FJSSP.py:70-72
        } else if (a0 == 3) { c = anyfav ? fav[s] : av[s]; kv = sortable(p.de[s]); }                         // :284-288
        else if (a0 == 4) { c = anyfav ? fav[s] : av[s]; kv = sortable_min_i32(e.due[s]); }                  // :289-293
        else { c = av[s]; kv = 0; }                                                                           // :294-295
        C[s] = c && run;
        key[s] = C[s] ? kv : 0ull;
        best = max(best, key[s]);
    }
    const uint64_t ext = gmax64(best);
    bool hit[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) hit[s] = C[s] && key[s] == ext;
    if (wave_any(run && a0 == 5)) {
        // random.choice(task_available_list): the idx-th available type in kind_task_tuple order
        const uint32_t m0 = gballot(av[0], e.gb), m1 = gballot(av[1], e.gb), m2 = gballot(av[2], e.gb), m3 = gballot(av[3], e.gb);
        const uint64_t avm = (uint64_t)(m0 | (m1 << 16)) | ((uint64_t)(m2 | (m3 << 16)) << 32);
        const bool rnd = run && a0 == 5;
        const int idx = g_rng(e, rnd, __builtin_popcountll(avm));
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            const int k = 16 * s + e.l;
            const int rank = __builtin_popcountll(avm & ((1ull << k) - 1ull));
            if (rnd) hit[s] = av[s] && rank == idx;
        }
    }
    int kc = kNone;
#pragma unroll
    for (int s = GS - 1; s >= 0; --s) kc = hit[s] ? 16 * s + e.l : kc;
    const int k = gmin(kc);
    return (run && k != kNone) ? k : -1;
}

// Strictly sequential sums out of the LDS rows: lanes with `walk` sum n8w entries of their row (rows are zero padded to
// n8w: +0.0 is an exact identity of a running sum that starts at +0.0), all of them side by side.
template <int V>
GDEV double g_walk(const GE<V> &e, bool walk, int row) {
    double sm = 0.0;
    if (wave_any(walk)) sm = lds_chain_sum_ring8(e.rows + (walk ? row : 0) * e.ks, e.n8w);
    return sm;
}

// Machine.gap_ave (class_FJSSP.py:144-146) of the machines in `cand` for the rows with `need`: a strictly sequential
// sum over kind_task_tuple order of unprocessed - fluid_unprocessed of the machine's operation types, divided by
// (n + 1e-18).  Lane (s, l) writes the gap of its operation type on every candidate machine to that machine's LDS row
// (+0.0 where the type cannot run there: an exact identity of the running sum); machine lane m walks row m.  Returns
// the machine lanes' gap_ave.
template <int V>
GDEV double g_gap_ave(const GE<V> &e, const DevBatch &b, bool need, uint32_t cand) {
    const double dt = (double)e.t;
    const double *col = reinterpret_cast<const double *>(e.ir + b.L.i_col);
    const int MP = e.MP;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (s < e.nslots) {
            const int k = 16 * s + e.l;
            const uint32_t elig = e.em[s] & 0xFFu;
            const uint32_t asg = (e.asgw >> (8 * s)) & 0xFFu;
            for (int m = 0; m < MP; ++m) {
                const bool row_on = need && ((cand >> m) & 1u);
                const bool on = row_on && ((elig >> m) & 1u);
                double g = 0.0;
                if (on) {
                    const double2 ar = *reinterpret_cast<const double2 *>(col + 2 * (k * MP + m));
                    // (unprocessed = arrival, less one where this type was assigned to m: class_FJSSP.py:198, :304)
                    const double un = asg == (uint32_t)m ? ar.x - 1.0 : ar.x;
                    g = un - (ar.x - dt * ar.y);
                }
                if (row_on && k < e.n8w) e.rows[m * e.ks + k] = g;      // (rows are n8w entries long: no spill into the next row)
            }
        }
    }
    lds_sync();
    const bool walk = need && e.l < e.M && ((cand >> e.l) & 1u);
    const double sm = g_walk<V>(e, walk, e.l);
    lds_sync();
    return sm / ((double)e.mcnt + 1e-18);
}

template <class T>
GDEV T pick_slot(const T (&a)[GS], int s) {
    const T lo = (s & 1) ? a[1] : a[0], hi = (s & 1) ? a[3] : a[2];
    return (s & 2) ? hi : lo;
}

// SO_FJSSP.py:300-322 / MO_FJSSP_discretes.py:209-230 machine_select for the rows with `go`.  Machine ids < 8: every
// CPython set involved iterates in ascending order (fjsp_pyset.h), the candidate lists are bit masks and "first
// extremum wins" is the lowest machine lane that attains it.  Returns m or -1 (status set); *pm_out its processing time.
template <int V>
GDEV int g_machine_select(GE<V> &e, const DevBatch &b, bool go, int a1, int k_sel, uint32_t em_sel, uint32_t idle, int *pm_out) {
    const uint32_t sel = idle & em_sel & 0xFFu, fsel = idle & (em_sel >> 8) & 0xFFu;
    const bool noev = go && sel == 0;
    if (noev) e.status |= FJSP_ST_NO_EVENT;
    constexpr bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES;
    const bool bad = go && sel != 0 && a1 >= (is_mo ? 3 : 5);
    if (bad) e.status |= FJSP_ST_BAD_MACHINE_RULE;                        // MyError :321
    const bool run = go && sel != 0 && !bad;
    // lane m: gap_rj_dict[m][k_sel] (class_FJSSP.py:137-142) and p[m][k_sel]; op-major layout: the column of k_sel is MP
    // contiguous entries per array
    const uint32_t fl = fsel ? fsel : sel;
    enum { GAP, PT, GAVE, RND };
    int mode;
    uint32_t C;
    if (is_mo) {
        mode = a1 == 0 ? (fsel ? GAP : PT) : (a1 == 1 ? GAVE : GAP);                                           // :213-227
        C = a1 == 0 ? (fsel ? fsel : sel) : fl;
    } else {
        mode = a1 <= 1 ? GAP : (a1 == 2 ? PT : (a1 == 3 ? GAVE : RND));                                         // :304-319
        C = (a1 == 0 || a1 == 3) ? fl : sel;
    }
    if (!run) C = 0;
    const bool mem = e.l < 8 && ((C >> e.l) & 1u);
    int pm = 0;
    double g = 0.0;
    if (run && e.l < 8 && ((sel >> e.l) & 1u)) {
        const int o = k_sel * e.MP + e.l;
        pm = reinterpret_cast<const uint16_t *>(e.ir + b.L.i_p)[o];
        const double2 ar = *reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(e.ir + b.L.i_col) + 2 * o);
        // (k_sel has not been dispatched yet -- it is available -- so its unprocessed is its arrival)
        g = ar.x - (ar.x - (double)e.t * ar.y);
    }
    uint64_t key = mode == PT ? sortable_min_i32(pm) : sortable(g);
    const bool need3 = run && mode == GAVE && (C & (C - 1)) != 0;         // (a list of one is returned without ranking it)
    if (wave_any(need3)) {
        const double gave = g_gap_ave<V>(e, b, need3, C);
        if (need3) key = sortable(gave);
    }
    if (mode == GAVE && !need3) key = 0;
    if (!mem) key = 0;
    const uint64_t ext = gmax64(key);
    bool hit = mem && key == ext;
    if (wave_any(run && mode == RND)) {
        const bool rnd = run && mode == RND;
        const int idx = g_rng(e, rnd, __builtin_popcount(sel));
        if (rnd) hit = mem && __builtin_popcount(C & ((1u << e.l) - 1u)) == idx;                                 // :318-319
    }
    const int m = gmin(hit ? e.l : kNone);
    const bool ok = run && m != kNone;
    *pm_out = gread(pm, ok ? m : 0, e.gb);
    return ok ? m : -1;
}

// SO_FJSSP.py:176-250: dispatch the job of k_sel on m_sel, then advance the clock until some operation type is
// available again (or the episode ends), for the rows with `go`.
template <int V>
GDEV void g_dispatch_advance(GE<V> &e, bool go, int k_sel, int m_sel, int pm, uint32_t kb_sel, int due_sel) {
    const int j_sel = (int)(kb_sel & 0xFFu), Jr = (int)((kb_sel >> 8) & 0xFFu), r_sel = (int)((kb_sel >> 16) & 0xFFu);
    const int time_end = e.t + pm;                                           // :184
    const int nj = j_sel + 1;                // the job of (r, j) is at stage j; it moves to j + 1
    const uint32_t word = jst_pack(kNoSeq, (uint32_t)nj);
    if (go && e.l == r_sel) e.jwl = word;                                    // :186-191
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (go && (int)((e.kB[s] >> 16) & 0xFFu) == r_sel) e.jw[s] = word;
        if (go && k_sel == 16 * s + e.l) e.asgw = (e.asgw & ~(0xFFu << (8 * s))) | ((uint32_t)m_sel << (8 * s));   // :198
    }
    // machine lane: time_end, and job | (k of the job's next stage + 1) << 16 (0 in the high half: no further stage)
    if (go && e.l == m_sel) { e.tend = time_end; e.mjob = r_sel | ((nj == Jr ? 0 : k_sel + 2) << 16); }            // :194-197
    if (go) {
        e.busy |= 1u << m_sel;
        e.completion = max(e.completion, time_end);
        if (nj == Jr) {                                                      // :200-202
            e.n_un--;
            const int late = time_end - due_sel;
            e.tard_done += late > 0 ? late : 0;
        }
    }
    bool act = go;
    for (;;) {
        const uint32_t idle = ~e.busy & e.mmask;
        bool av_l = false;
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            const uint32_t kb = e.kB[s], js = e.jw[s];
            const bool wait = ((kb >> 24) & 2u) && (js & 0xFFu) == (kb & 0xFFu) && (js >> 8) != kNoSeq;
            av_l = av_l || (wait && (e.em[s] & idle & 0xFFu) != 0);
        }
        act = act && gballot(av_l, e.gb) == 0;                               // :204
        if (!wave_any(act)) break;
        const int tn = gmin((e.l < e.M && e.tend > e.t) ? e.tend : 0x7fffffff);   // :205-207 next event
        if (act && tn == 0x7fffffff) { e.status |= FJSP_ST_NO_EVENT; act = false; }
        if (act) e.t = tn;
        // :209-215 completions in ascending machine order: the job goes to the FIFO of its next stage (if any)
        const bool rel = act && e.l < e.M && e.tend == tn && (e.mjob >> 16) != 0;
        const uint32_t relm = gballot(rel, e.gb);
        const int rank = __builtin_popcount(relm & ((1u << e.l) - 1u));
        // every releasing machine lane hands its job lane the job's place in the append order (ds_permute: a push; the
        // other lanes push a zero to lane 15, which is no job lane -- at most 15 jobs)
        const int got = __builtin_amdgcn_ds_permute((e.gb + (rel ? (e.mjob & 0xFFFF) : 15)) << 2, rel ? rank + 1 : 0);
        if (got != 0 && e.l < 15) e.jwl = (e.jwl & 0xFFu) | ((e.seq_ctr + (uint32_t)got - 1u) << 8);
        e.seq_ctr += (uint32_t)__builtin_popcount(relm);
        g_refresh_jw<V>(e);
        const uint32_t freed = gballot(e.l < e.M && e.tend <= e.t, e.gb);   // :233-235
        if (act) {
            e.busy &= ~freed;
            if (e.n_un == 0) { e.done = 1; act = false; }                    // :247-250
        }
    }
}

// SO_FJSSP.py:78-97 state_extract (+ the ratios of update_parameter :156-165) of the rows with `on` at the current
// clock; g_params must be current.  Returns the observation in the observation lanes (lane (8 + i) & 15 holds entry i)
// and delay_time_sum_unprocessed (:110-122) through *tard_unproc.  stats_only: a step that hands no state back needs
// the tardiness for its reward and nothing else.
template <int V>
GDEV double g_observe(const GE<V> &e, const GP &p, bool on, bool stats_only, long long *tard_unproc) {
    using P = ObsPos<V>;
    constexpr int L_ave0 = (P::ave0 + 8) & 15, L_ave1 = (P::ave1 + 8) & 15, L_ave2 = (P::ave2 + 8) & 15;
    constexpr int L_sd0 = (P::sd0 + 8) & 15, L_sd1 = (P::sd1 + 8) & 15, L_sd2 = (P::sd2 + 8) & 15;
    // ---- integer statistics (order-free): packed row sums
    uint32_t c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (s < e.nslots) {
            const bool last = ((e.kB[s] >> 24) & 1u) != 0;
            const uint32_t tl = (last && p.late[s]) ? (uint32_t)p.da[s] : 0u;                 // :120-122
            c1 += (p.in[s] ? 1u : 0u) | (p.late[s] ? 1u << 8 : 0u) | (p.cnte[s] ? 1u << 16 : 0u) | ((last && p.late[s]) ? 1u << 24 : 0u);
            c2 += ((last && p.cnte[s]) ? 1u : 0u) | ((tl & 0xFFFFu) << 8);
            c3 += tl >> 16;
        }
    }
    c1 = (uint32_t)gsum((int)c1); c2 = (uint32_t)gsum((int)c2); c3 = (uint32_t)gsum((int)c3);
    const int tasks = (int)(c1 & 0xFFu), delay_a = (int)((c1 >> 8) & 0xFFu), delay_e = (int)((c1 >> 16) & 0xFFu), job_a = (int)(c1 >> 24);
    const int job_e = (int)(c2 & 0xFFu);
    *tard_unproc = (long long)(c2 >> 8) + ((long long)c3 << 16);
    if (stats_only) return 0.0;
    const int jobs = e.n_un;
    // ---- first pass: the three means.  finish_rate is 0 or 1 here (one job per kind) and the machines' time_end are
    // integers: their left-to-right f64 sums are exact, i.e. equal to the integer sums; gap_rate's is walked.
    const double dt = (double)e.t;
    double frv[GS], grv[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (s < e.nslots) {
            const bool valid = ((e.kB[s] >> 24) & 2u) != 0;
            const double fq = 1.0 - e.rsum[s] * dt;                               // fluid_unprocessed_number, Q0 = 1 (:239-240)
            frv[s] = (valid && !p.in[s]) ? 1.0 : 0.0;                             // finish_rate class_FJSSP.py:74-76 ((1 - nun) / 1)
            grv[s] = valid ? (p.in[s] ? 1.0 : 0.0) - fq : 0.0;                    // gap_rate class_FJSSP.py:66-68 (x / 1.0 == x)
            if (on && 16 * s + e.l < e.n8w) e.rows[16 * s + e.l] = grv[s];
        } else { frv[s] = 0.0; grv[s] = 0.0; }
    }
    const uint32_t tlo = (uint32_t)gsum((e.l < e.M) ? (e.tend & 0xFFFF) : 0), thi = (uint32_t)gsum((e.l < e.M) ? (int)((uint32_t)e.tend >> 16) : 0);
    const double tsum_td = (double)(((long long)thi << 16) + (long long)tlo);
    lds_sync();
    const double cs1 = g_walk<V>(e, on && e.l == L_ave1, 0);
    lds_sync();
    double num = 0.0, den = 1.0;
    if (e.l == L_ave0) { num = (double)(e.K - tasks); den = (double)e.K; }
    if (e.l == L_ave1) { num = cs1; den = (double)e.K; }
    if (e.l == L_ave2) { num = tsum_td; den = (double)e.M; }
    {
        const int q = ((e.l - 8) & 15) - P::ratio0;                               // :156-165
        if (q >= 0 && q < 4) {
            num = (double)(q == 0 ? delay_a : (q == 1 ? delay_e : (q == 2 ? job_a : job_e)));
            den = (double)(q < 2 ? tasks : jobs);
        }
    }
    const double q1 = num / den;
    const double ave_fr = bcd<L_ave0>(q1), ave_gr = bcd<L_ave1>(q1), ave_td = bcd<L_ave2>(q1);
    // ---- second pass: squared deviations (math.pow(d, 2), :86-95), population standard deviations
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (s < e.nslots) {
            const bool valid = ((e.kB[s] >> 24) & 2u) != 0;
            const double d1 = frv[s] - ave_fr, d2 = grv[s] - ave_gr;
            if (on && 16 * s + e.l < e.n8w) { e.rows[16 * s + e.l] = valid ? d1 * d1 : 0.0; e.rows[e.ks + 16 * s + e.l] = valid ? d2 * d2 : 0.0; }
        }
    }
    double d3 = (double)e.tend - ave_td;
    d3 = e.l < e.M ? d3 * d3 : 0.0;
    double cs_td = 0.0;      // machines in ascending order (+0.0 beyond M)
    cs_td = cs_td + bcd<0>(d3); cs_td = cs_td + bcd<1>(d3); cs_td = cs_td + bcd<2>(d3); cs_td = cs_td + bcd<3>(d3);
    cs_td = cs_td + bcd<4>(d3); cs_td = cs_td + bcd<5>(d3); cs_td = cs_td + bcd<6>(d3); cs_td = cs_td + bcd<7>(d3);
    lds_sync();
    const double cs2 = g_walk<V>(e, on && (e.l == L_sd0 || e.l == L_sd1), e.l == L_sd1 ? 1 : 0);
    lds_sync();
    double num2 = 0.0, den2 = 1.0;
    if (e.l == L_sd0 || e.l == L_sd1) { num2 = cs2; den2 = (double)e.K; }
    if (e.l == L_sd2) { num2 = cs_td; den2 = (double)e.M; }
    const double q2 = sqrt(num2 / den2);
    const int oi = (e.l - 8) & 15;
    double cur = q1;
    if (oi >= P::ratio0 && oi < P::ratio0 + 4 && e.done) cur = 0.0;               // the ratios are 0 once the episode is over
    if (e.l == L_sd0 || e.l == L_sd1 || e.l == L_sd2) cur = q2;
    if (is_so_v<V> && oi == 0) cur = (double)e.M;
    return cur;
}

// Reward and bookkeeping of step() (SO_FJSSP.py:259-265, MO_FJSSP_discretes.py:232-244) once the observation is out.
struct MoW { double w0, w1, cn, tn; };      // MO_FJSSP_discretes.py:88 weight vector + normalisers (defaults: d_mo == NULL)
template <int V>
GDEV double g_reward(GE<V> &e, bool go, const MoW &mo, long long tard_unproc) {
    const long long delay_new = e.tard_done + tard_unproc;                       // :259
    const long long delta = delay_new - e.delay_sum;
    const int dc = e.completion_last - e.completion;
    if (go) { e.delay_sum = delay_new; e.completion_last = e.completion; }       // :263
    double r;
    if (V != FJSP_VARIANT_MO_FJSSP_DISCRETES) r = (double)(-delta);              // :328 (exact integer)
    else {
        const double w0 = mo.w0, w1 = mo.w1, cn = mo.cn, tn = mo.tn;
        if (cn > 0.0 && tn > 0.0) r = (double)dc / cn * w0 + (double)(-delta) / tn * w1;
        else if (w1 == 1.0) r = (double)(-delta);
        else if (w0 == 1.0) r = (double)dc;
        else { r = 0.0; if (go) e.status |= FJSP_ST_BAD_TASK_RULE; }             // MyError :244
    }
    return go ? r : 0.0;
}

// state = [static, v(t), v(t) - v(t-1)]  (SO_FJSSP.py:71-72,257-258); updates obs_prev
template <int V>
GDEV void g_emit(GE<V> &e, bool on, double cur, const double *sstate, double *state_out) {
    constexpr int n_obs = kNObs<V>, n_static = kNStatic<V>;
    const int oi = (e.l - 8) & 15;
    const double gap = cur - e.obs_prev;
    if (on && oi < n_obs) {
        e.obs_prev = cur;
        if (state_out) {
            double *o = state_out + (size_t)e.env * (n_static + 2 * n_obs);
            o[n_static + oi] = cur; o[n_static + n_obs + oi] = gap;
        }
    }
    if (n_static > 0 && state_out && on && e.l < n_static) state_out[(size_t)e.env * (n_static + 2 * n_obs) + e.l] = sstate[e.l];
}

// Bind the four rows of a wave to their records and bring the environments in: every load is independent of the
// others (bounds come from the kernel arguments), one memory round trip.
template <int V>
GDEV void g_open(GE<V> &e, const DevBatch &b, int wave_id, unsigned char *lds) {
    using FO = FixedOffsets;
    const int lane = (int)__lane_id();
    e.l = lane & 15; e.gb = lane & 48;
    const int env_raw = wave_id * 4 + (lane >> 4);
    e.live = env_raw < b.N;
    e.env = min(env_raw, b.N - 1);
    const int MP = b.MP, JP = b.JP;
    e.MP = MP;
    const int inst = b.n_inst == b.N ? e.env : e.env % b.n_inst;
    const unsigned char *ir = b.inst + (size_t)inst * b.L.i_stride;
    unsigned char *er = b.envs + (size_t)e.env * FO::e_stride_plain((uint32_t)MP, (uint32_t)JP, 64u, true);
    e.ir = ir; e.er = er;
    e.ks = row_stride(b.kmax);
    e.rows = reinterpret_cast<double *>(lds) + (size_t)(lane >> 4) * (MP < 2 ? 2 : MP) * e.ks;
    // ---- issue every load
    const int4 h = *reinterpret_cast<const int4 *>(ir);
    uint4 A[GS];
    double2 B[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        A[s] = make_uint4(0u, 0u, 0u, 0u); B[s] = make_double2(0.0, 0.0);
        if (16 * s + e.l < b.kmax) {
            const unsigned char *slot = ir + b.L.i_op + s * 512;
            A[s] = reinterpret_cast<const uint4 *>(slot)[e.l];
            B[s] = reinterpret_cast<const double2 *>(slot + 256)[e.l];
        }
    }
    const uint32_t asgw = reinterpret_cast<const uint32_t *>(er + FO::e_asg((uint32_t)MP, (uint32_t)JP, 64u, true))[e.l];
    uint32_t jwl = 0;
    if (e.l < b.jcap) jwl = reinterpret_cast<const uint32_t *>(er + FO::e_jst((uint32_t)MP))[e.l];
    int tend = 0, mjob = -1;
    if (e.l < MP) {
        tend = reinterpret_cast<const int32_t *>(er + FO::e_tend())[e.l];
        mjob = reinterpret_cast<const int32_t *>(er + FO::e_mjob((uint32_t)MP))[e.l];
    }
    const long long sc = reinterpret_cast<const long long *>(er)[e.l];            // EnvScalars words 0..15
    long long sc2 = 0;
    if (e.l < 2) sc2 = reinterpret_cast<const long long *>(er)[16 + e.l];         // obs_prev[8], [9]
    // ---- consume
    e.K = h.x; e.M = h.y; e.njobs = h.w;
    e.mmask = (1u << e.M) - 1u;
    {
        const int k0 = __builtin_amdgcn_readlane(e.K, 0), k1 = __builtin_amdgcn_readlane(e.K, 16);
        const int k2 = __builtin_amdgcn_readlane(e.K, 32), k3 = __builtin_amdgcn_readlane(e.K, 48);
        const int kw = max(max(k0, k1), max(k2, k3));
        e.nslots = (kw + 15) >> 4;
        e.n8w = max((kw + 7) & ~7, 8);
    }
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        e.kB[s] = A[s].x; e.em[s] = A[s].y; e.due[s] = (int)A[s].z;
        e.rsum[s] = B[s].x; e.tsum[s] = B[s].y;
    }
    e.mcnt = (int)A[0].w;                                   // (slot 0, lane m: operation types machine m can process)
    e.asgw = asgw; e.jwl = jwl;
    e.tend = e.l < e.M ? tend : 0; e.mjob = e.l < e.M ? mjob : -1;
    const int lo = (int)sc, hi = (int)(sc >> 32);
    e.t = bc<0>(lo); e.step_count = bc<0>(hi);
    e.done = bc<1>(lo); e.n_un = bc<1>(hi);
    e.status = bcu<2>((uint32_t)lo); e.seq_ctr = bcu<2>((uint32_t)hi);
    e.rng_calls = bcu<3>((uint32_t)lo); e.busy = bcu<3>((uint32_t)hi);
    e.completion = bc<4>(lo); e.completion_last = bc<4>(hi);
    e.tard_done = bcl<5>(sc); e.delay_sum = bcl<6>(sc);
    e.t_arr = bc<7>(lo); e.misc = bcu<7>((uint32_t)hi);
    e.obs_prev = __longlong_as_double(e.l >= 8 ? sc : sc2);
    e.env_seed = b.rng_seed + (uint64_t)e.env * 1000003ULL;
    g_refresh_jw<V>(e);
}

template <int V>
GDEV void g_store(const GE<V> &e, const DevBatch &b) {
    using FO = FixedOffsets;
    if (!e.live) return;
    const int MP = e.MP, JP = b.JP;
    unsigned char *er = e.er;
    // EnvScalars words 0..7 from the row-uniform values, 8..17 = obs_prev.  (The values pass through empty asm statements:
    // a select between loads of struct fields would be rewritten into a load through a selected ADDRESS, which pins the
    // whole row state in scratch.)
    int v_t = e.t, v_sc = e.step_count, v_done = e.done, v_nun = e.n_un, v_st = (int)e.status, v_seq = (int)e.seq_ctr;
    int v_rng = (int)e.rng_calls, v_busy = (int)e.busy, v_c = e.completion, v_cl = e.completion_last;
    int v_tdl = (int)(uint32_t)e.tard_done, v_tdh = (int)(e.tard_done >> 32), v_dsl = (int)(uint32_t)e.delay_sum, v_dsh = (int)(e.delay_sum >> 32);
    int v_ta = e.t_arr, v_misc = (int)e.misc;
    asm("" : "+v"(v_t), "+v"(v_sc), "+v"(v_done), "+v"(v_nun), "+v"(v_st), "+v"(v_seq), "+v"(v_rng), "+v"(v_busy));
    asm("" : "+v"(v_c), "+v"(v_cl), "+v"(v_tdl), "+v"(v_tdh), "+v"(v_dsl), "+v"(v_dsh), "+v"(v_ta), "+v"(v_misc));
    int lo = v_t, hi = v_sc;
    if (e.l == 1) { lo = v_done; hi = v_nun; }
    if (e.l == 2) { lo = v_st; hi = v_seq; }
    if (e.l == 3) { lo = v_rng; hi = v_busy; }
    if (e.l == 4) { lo = v_c; hi = v_cl; }
    if (e.l == 5) { lo = v_tdl; hi = v_tdh; }
    if (e.l == 6) { lo = v_dsl; hi = v_dsh; }
    if (e.l == 7) { lo = v_ta; hi = v_misc; }
    long long w = (long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    if (e.l >= 8) w = __double_as_longlong(e.obs_prev);
    reinterpret_cast<long long *>(er)[e.l] = w;
    if (e.l < 2) reinterpret_cast<long long *>(er)[16 + e.l] = __double_as_longlong(e.obs_prev);
    if (e.l < e.M) {
        reinterpret_cast<int32_t *>(er + FO::e_tend())[e.l] = e.tend;
        reinterpret_cast<int32_t *>(er + FO::e_mjob((uint32_t)MP))[e.l] = e.mjob;
    }
    if (e.l < e.njobs) reinterpret_cast<uint32_t *>(er + FO::e_jst((uint32_t)MP))[e.l] = e.jwl;
    reinterpret_cast<uint32_t *>(er + FO::e_asg((uint32_t)MP, (uint32_t)JP, 64u, true))[e.l] = e.asgw;
}

// SO_FJSSP.py:51-76 reset of the rows with `on`, as the autoreset path of a step needs it (fresh-object semantics; the
// observation of the reset state was published per instance by reset_kernel: Layout::i_obs0); rng_calls survives
template <int V>
GDEV void g_restart(GE<V> &e, const DevBatch &b, bool on) {
    const int oi = (e.l - 8) & 15;
    double o0 = 0.0;
    if (on && oi < kNObs<V>) o0 = reinterpret_cast<const double *>(e.ir + b.L.i_obs0)[oi];
    if (on) {
        e.t = 0; e.step_count = 0; e.done = 0; e.status = 0; e.busy = 0; e.completion = 0; e.completion_last = 0;
        e.tard_done = 0; e.delay_sum = 0; e.tend = 0; e.mjob = -1;
        e.t_arr = 0; e.misc = 1u;                                               // next_order 1, nothing pending, observation current
        e.n_un = e.njobs; e.seq_ctr = (uint32_t)e.njobs;
        if (e.l < e.njobs) e.jwl = jst_pack((uint32_t)e.l, 0u);               // class_FJSSP.py:225 (job n = kind n)
        e.asgw = 0xFFFFFFFFu;                                                   // :304 unprocessed = arrival
        e.obs_prev = o0;
    }
    g_refresh_jw<V>(e);
}

// One step() of the rows with go_in.  need_obs: the caller wants the state vector (state_out may still be null: the
// policy rollout keeps it in registers).  Returns the reward; *k_out / *m_out the chosen pair (-1: none).
template <int V>
GDEV double g_step(GE<V> &e, const DevBatch &b, bool go_in, int a0, int a1, const MoW &mo, bool need_obs, double *state_out,
                   int *k_out, int *m_out) {
    constexpr bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES;
    bool go = go_in;
    GP p;
    g_params<V>(e, p);
    const bool stale = (e.misc >> 24) & 1u;
    if (need_obs && wave_any(go && stale)) {
        // v(t-1) of the state vector is the observation of the environment as this step finds it; steps that handed no
        // state back did not keep it current (fjsp_kernels.hip obs_refresh)
        long long tu;
        const double v = g_observe<V>(e, p, go && stale, false, &tu);
        if (go && stale && ((e.l - 8) & 15) < kNObs<V>) e.obs_prev = v;
        if (go && stale) e.misc &= ~(1u << 24);
    }
    if (is_mo) {                     // flat action -> self.actions[action] (MO_FJSSP_discretes.py:26,92)
        if (go && a0 >= 18) { e.status |= FJSP_ST_BAD_TASK_RULE; go = false; }   // IndexError
        a1 = a0 % 3; a0 = a0 / 3;
    }
    const uint32_t idle = ~e.busy & e.mmask;
    const int k_sel = g_task_select<V>(e, p, go, a0, idle);
    go = go && k_sel >= 0;
    const int ks = go ? k_sel : 0;
    const uint32_t kb_sel = greadu(pick_slot(e.kB, ks >> 4), ks & 15, e.gb);
    const uint32_t em_sel = greadu(pick_slot(e.em, ks >> 4), ks & 15, e.gb);
    const int due_sel = gread(pick_slot(e.due, ks >> 4), ks & 15, e.gb);
    int pm = 0;
    const int m_sel = g_machine_select<V>(e, b, go, a1, ks, em_sel, idle, &pm);
    *k_out = k_sel; *m_out = m_sel;
    go = go && m_sel >= 0;
    g_dispatch_advance<V>(e, go, ks, m_sel, pm, kb_sel, due_sel);
    if (go) e.step_count++;                                                      // :252
    g_params<V>(e, p);
    long long tard_unproc = 0;
    const double cur = g_observe<V>(e, p, go, !need_obs, &tard_unproc);       // :256
    if (need_obs) g_emit<V>(e, go, cur, reinterpret_cast<const double *>(e.ir + b.L.i_ss), state_out);
    else if (go) e.misc |= 1u << 24;
    return g_reward<V>(e, go, mo, tard_unproc);
}

// ------------------------------------------------------------------------ kernels
// One step of every environment of a group batch (fjsp_kernels.hip step_kernel for the same batch gives the same results)
template <int V>
__global__ __launch_bounds__(64) void gstep_kernel(DevBatch b, const uint8_t *actions, const double *mo, int autoreset, double *state_out,
                                                   double *reward_out, uint8_t *done_out, int16_t *trace_km) {
    GE<V> e;
    const int wave_id = (int)blockIdx.x;
    // the action pair of the row's environment (2-byte aligned: checked by the host entry points)
    const int env0 = min(wave_id * 4 + (int)(__lane_id() >> 4), b.N - 1);
    const uint32_t araw = reinterpret_cast<const uint16_t *>(actions)[env0];
    MoW mw = {0.0, 1.0, 0.0, 0.0};
    if (V == FJSP_VARIANT_MO_FJSSP_DISCRETES && mo) {
        const double2 m01 = *reinterpret_cast<const double2 *>(mo + (size_t)env0 * 4), m23 = *reinterpret_cast<const double2 *>(mo + (size_t)env0 * 4 + 2);
        mw.w0 = m01.x; mw.w1 = m01.y; mw.cn = m23.x; mw.tn = m23.y;
    }
    g_open<V>(e, b, wave_id, g_lds);
    const int a0 = (int)(araw & 0xFFu), a1 = (int)(araw >> 8);
    bool go = e.live;
    if (wave_any(go && e.done != 0)) {
        const bool was_done = go && e.done != 0;
        if (autoreset == 1) g_restart<V>(e, b, was_done);
        else {
            if (was_done && autoreset == 0) e.status |= FJSP_ST_STEP_AFTER_DONE;      // 2: idle silently
            go = go && !was_done;
        }
    }
    int k_sel = -1, m_sel = -1;
    const double reward = g_step<V>(e, b, go, a0, a1, mw, state_out != nullptr, state_out, &k_sel, &m_sel);
    if (e.live && e.l == 0) {
        if (reward_out) reward_out[e.env] = reward;
        if (done_out) done_out[e.env] = (uint8_t)e.done;
        if (trace_km) { trace_km[(size_t)e.env * 2] = (int16_t)k_sel; trace_km[(size_t)e.env * 2 + 1] = (int16_t)m_sel; }
    }
    g_store<V>(e, b);
}

// T fused steps per launch with the actions given (rule sweeps): the environments live in registers for the whole
// launch.  Same outputs as fjsp_kernels.hip rollout_kernel.
template <int V>
__global__ __launch_bounds__(64) void grollout_kernel(DevBatch b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km,
                                                      double *reward_out, double *state_last) {
    GE<V> e;
    const int wave_id = (int)blockIdx.x;
    g_open<V>(e, b, wave_id, g_lds);
    MoW mw = {0.0, 1.0, 0.0, 0.0};
    if (V == FJSP_VARIANT_MO_FJSSP_DISCRETES && mo) {
        mw.w0 = mo[(size_t)e.env * 4]; mw.w1 = mo[(size_t)e.env * 4 + 1]; mw.cn = mo[(size_t)e.env * 4 + 2]; mw.tn = mo[(size_t)e.env * 4 + 3];
    }
    for (int s = 0; s < T; ++s) {
        const size_t o = (size_t)s * b.N + e.env;
        const bool go = e.live && !e.done && !(e.status & (FJSP_ST_BAD_TASK_RULE | FJSP_ST_BAD_MACHINE_RULE | FJSP_ST_NO_EVENT));
        if (!wave_any(go)) {
            if (e.live && e.l == 0) {
                if (trace_km) { trace_km[o * 2] = -1; trace_km[o * 2 + 1] = -1; }
                if (reward_out) reward_out[o] = 0.0;
            }
            continue;
        }
        const uint32_t araw = reinterpret_cast<const uint16_t *>(actions)[o];
        int k_sel = -1, m_sel = -1;
        const double reward = g_step<V>(e, b, go, (int)(araw & 0xFFu), (int)(araw >> 8), mw, state_last != nullptr,
                                        state_last, &k_sel, &m_sel);
        if (e.live && e.l == 0) {
            if (trace_km) { trace_km[o * 2] = (int16_t)k_sel; trace_km[o * 2 + 1] = (int16_t)m_sel; }
            if (reward_out) reward_out[o] = reward;
        }
    }
    g_store<V>(e, b);
}

}  // namespace grp

// ------------------------------------------------------------------ host launchers
template <class K>
static inline void grp_allow_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

int launch_step_group(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                      uint8_t *done, int16_t *trace_km, hipStream_t st) {
    if (!b.grp) return -1;
    const size_t lds = grp::group_lds_bytes(b.kmax, b.MP);
    const dim3 grid((unsigned)((b.N + 3) / 4));
    if (b.variant == FJSP_VARIANT_SO_FJSSP)
        hipLaunchKernelGGL((grp::gstep_kernel<FJSP_VARIANT_SO_FJSSP>), grid, dim3(64), lds, st, b, actions, mo, autoreset, state, reward, done, trace_km);
    else if (b.variant == FJSP_VARIANT_MO_FJSSP_DISCRETES)
        hipLaunchKernelGGL((grp::gstep_kernel<FJSP_VARIANT_MO_FJSSP_DISCRETES>), grid, dim3(64), lds, st, b, actions, mo, autoreset, state, reward, done, trace_km);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_rollout_group(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                         double *state_last, hipStream_t st) {
    if (!b.grp) return -1;
    const size_t lds = grp::group_lds_bytes(b.kmax, b.MP);
    const dim3 grid((unsigned)((b.N + 3) / 4));
    if (b.variant == FJSP_VARIANT_SO_FJSSP)
        hipLaunchKernelGGL((grp::grollout_kernel<FJSP_VARIANT_SO_FJSSP>), grid, dim3(64), lds, st, b, actions, mo, T, trace_km, reward, state_last);
    else if (b.variant == FJSP_VARIANT_MO_FJSSP_DISCRETES)
        hipLaunchKernelGGL((grp::grollout_kernel<FJSP_VARIANT_MO_FJSSP_DISCRETES>), grid, dim3(64), lds, st, b, actions, mo, T, trace_km, reward, state_last);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace fjsp
