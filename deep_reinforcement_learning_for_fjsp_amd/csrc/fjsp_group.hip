// gfx950 (MI355X / CDNA4) kernels of the batched rule-dispatch FJSP environment, second family:
// ONE 16-LANE DPP ROW PER ENVIRONMENT, four environments per wavefront.
//
// fjsp_kernels.hip gives every environment a whole wavefront; on the 10x5 / Brandimarte workloads (one job per kind,
// K <= 64 operation types, M <= 8 machines, <= 15 jobs) 40 of its 64 lanes carry an operation type, 5 a machine, and the
// wave-uniform part of a step (the scalar stream, the reductions, the serial sums) is paid per environment.  Here
//   * lane l of row g owns the operation types k = 16 s + l of environment g, s = 0..3 ("slots": registers);
//   * lanes 0..njobs-1 of the row double as its JOB lanes (state word of job l and the data of the job's CURRENT
//     operation type), lanes 0..M-1 as its machine lanes, lane (8 + i) & 15 keeps observation i;
//   * with one job per kind only the current operation of a waiting job can be available, and kind_task_tuple order
//     among those is job order: task_select, the event loop's "is anything available" and update_parameter's per-job
//     statistics run on the job lanes (one register per row instead of one per operation type); only the estimated-
//     delay counts, the observation's operand rows and Machine.gap_ave walk the operation slots;
//   * what was wave-uniform becomes row-uniform and lives in VGPRs: a value is read from a row's lane n with one DPP
//     move (row_newbcast:n), from a computed lane with ds_bpermute_b32; reductions are four DPP steps inside the row
//     (quad_perm, quad_perm, row_half_mirror, row_mirror) and leave the result in every lane of the row; a "first
//     extremum wins" choice is a row maximum of an order-preserving 64-bit key and the first set bit of the row's 16
//     bits of ballot(member && key == maximum);
//   * control flow is wave-uniform only ("does any row of this wave need ...", a ballot): inside, lanes are predicated;
//   * the strictly sequential f64 sums of the observation and of Machine.gap_ave are walked out of LDS rows by one
//     lane per sum, all rows of the wave side by side (the chains of four environments cost one instruction stream);
//   * the static data come from a packed copy of the instance rows (Layout::i_op: two 16-byte loads per slot and one
//     line of per-job / per-machine words), the assigned-machine bytes of a lane's four slots share one word (asg_pos());
//     the (operation x machine) rows Machine.gap_ave needs are requested together with the state when the row's rule
//     asks for them (they depend on nothing the step computes).
// The records in HBM are the ones of fjsp_kernels.hip: reset, the fused policy rollout, read-back and every other
// entry point keep working on the same batch, and a batch can be stepped by either family (FJSP_STEP_IMPL=wave).
//
// Reference restated (paths relative to the reference root), single-job form (every per-(r, j) list has at most one
// member, job index == kind index):
//   environments/SO_FJSSP.py:99-166   update_parameter -> g_task_select (keys), g_observe (statistics)
//   environments/SO_FJSSP.py:168-250  step, first half -> g_task_select, g_machine_select, g_dispatch_advance
//   environments/SO_FJSSP.py:267-322  task_select, machine_select (class_FJSSP.py:137-146 gap, gap_ave)
//   environments/SO_FJSSP.py:78-97, 252-265 state_extract, reward -> g_observe, g_reward
//   environments/MO_FJSSP_discretes.py:26,88-244 flat action, three machine rules, weighted reward
// Compiled with -ffp-contract=off: a*b+c must round twice like CPython.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/fjsp_amd.h"
#include "fjsp_common.h"
#include "fjsp_device.h"

#pragma clang fp contract(off)

namespace fjsp {
namespace grp {

constexpr int GS = 4;       // operation-type slots per lane
constexpr int KS = 66;      // doubles per LDS row: 16 GS + 2 (16-byte aligned rows; (2 KS) mod 64 = 4: the rows that are walked
                            // side by side start in different banks)
constexpr uint32_t kSeqNone = kNoSeq << 8;      // state words at or above it: the job is not waiting

#define GDEV __device__ __forceinline__
// slots 0..2 are always processed (an instance of at most 32 operation types wastes slot 2: rare shapes), slot 3 -- operation
// types 48..63 -- only when some row of the wave has that many: one wave-uniform branch per section instead of four
#define SLOT_ON(e, s) ((s) < 3 || (e).nslots > 3)

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// -DFJSP_GSTAMPS builds a DIAGNOSTIC library (never shipped, never benchmarked): lane 0 of every wave adds the s_memtime
// delta of each phase of a step to a global table (tools/stamp_group.py)
#ifdef FJSP_GSTAMPS
__device__ unsigned long long fjsp_gstamp_acc[16];
__device__ unsigned long long g_stamp_t0;
#define GSTAMP_DECL unsigned long long gst[14]; unsigned long long gst_t0
#define GSTAMP_BEGIN() do { for (int _i = 0; _i < 14; ++_i) gst[_i] = 0; gst_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define GSTAMP(slot) do { const unsigned long long _t1 = __builtin_amdgcn_s_memtime(); gst[slot] += _t1 - gst_t0; gst_t0 = _t1; } while (0)
#define GSTAMP_FLUSH() do { if (__lane_id() == 0) { for (int _i = 0; _i < 14; ++_i) atomicAdd(&fjsp_gstamp_acc[_i], gst[_i]); atomicAdd(&fjsp_gstamp_acc[15], 1ull); } } while (0)
#else
#define GSTAMP_DECL
#define GSTAMP_BEGIN()
#define GSTAMP(slot)
#define GSTAMP_FLUSH()
#endif

// ---- row primitives
template <int N>
GDEV int bc(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xF, 0xF, false); }       // row_newbcast:N (every lane is read from: `old` is never used)
template <int N>
GDEV uint32_t bcu(uint32_t v) { return (uint32_t)bc<N>((int)v); }
template <int N>
GDEV long long bcl(long long v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xF, 0xF, false); }
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
// a[s] for a per-lane slot index, written as selects between VALUES (an indexed load would pin the row state in scratch)
// (the values pass through an empty asm statement: a select between loads would be rewritten into a load through a selected
// ADDRESS)
GDEV uint32_t pick_slot(const uint32_t (&a)[GS], int s) {
    uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    const uint32_t lo = (s & 1) ? a1 : a0, hi = (s & 1) ? a3 : a2;
    return (s & 2) ? hi : lo;
}

// order-preserving 64-bit image of an f64 key (-0.0 folded onto +0.0: they compare equal in Python); keys are finite
GDEV uint64_t sortable(double z0) {
    const double z = z0 + 0.0;
    uint32_t hi = (uint32_t)__double2hiint(z), lo = (uint32_t)__double2loint(z);
    const uint32_t sg = (uint32_t)((int)hi >> 31);
    hi ^= sg | 0x80000000u; lo ^= sg;
    return ((uint64_t)hi << 32) | lo;
}
GDEV uint64_t sortable_max_i32(int v) { return (uint64_t)((uint32_t)v ^ 0x80000000u) << 32; }
GDEV uint64_t sortable_min_i32(int v) { return (uint64_t)(~((uint32_t)v ^ 0x80000000u)) << 32; }
// first member (lowest lane) of the row that attains the row maximum of `key`; 255 when the row has no member
GDEV int first_max(bool mem, uint64_t key, int gb) {
    const uint32_t hi = mem ? (uint32_t)(key >> 32) : 0u, lo = mem ? (uint32_t)key : 0u;
    const uint32_t mhi = gmaxu(hi);
    const uint32_t mlo = gmaxu(hi == mhi ? lo : 0u);      // the low words among the lanes that tie on the high word
    const uint32_t hits = gballot(mem && hi == mhi && lo == mlo, gb);
    return hits ? (int)__builtin_ctz(hits) : 255;
}

// LDS rows per environment: one per machine when Machine.gap_ave's operands are laid out for all of them at once (EARLY), three in
// the large-batch form (g_gap_ave_lean); the observation needs one
template <int MPC, bool EARLY>
__host__ __device__ constexpr int group_rows() { return EARLY ? MPC : 3; }
template <int MPC, bool EARLY>
__host__ __device__ constexpr size_t group_lds_bytes() { return (size_t)4 * group_rows<MPC, EARLY>() * KS * 8 + 256; }   // + the ring's read-ahead

// Resident copy of an instance's static tables in LDS (fused kernel of batches that leave a CU at most four waves): the
// fluid numbers of its operation types (B parts of the slots), the machine-major {arrival, rate} table and the processing
// times.  A step of the lean build fetches these from memory three times in a row (current operations' fluid numbers ->
// chosen operation's column -> the observation's fluid numbers); they never change during an episode.
template <int MPC>
__host__ __device__ constexpr uint32_t res_bytes() { return 2048u + (uint32_t)MPC * 1024u; }
GDEV const double2 *res_B(uint32_t res) { return reinterpret_cast<const double2 *>(g_lds + res); }                   // [64] {rate sum, time sum}
GDEV const uint16_t *res_P(uint32_t res) { return reinterpret_cast<const uint16_t *>(g_lds + res + 1024u); }         // [K][MP] as in memory (<= 1 KB)
GDEV const double2 *res_C(uint32_t res) { return reinterpret_cast<const double2 *>(g_lds + res + 2048u); }           // [MP][64] {arrival, rate}

// ------------------------------------------------------------------ row state
struct MoW { double w0, w1, cn, tn; };      // MO_FJSSP_discretes.py:88 weight vector + normalisers (defaults: d_mo == NULL)

template <int V>
struct GE {
    int l, gb, env;
    bool live;                 // the row has an environment (the last wave of a batch may not be full)
    const unsigned char *ir;
    unsigned char *er;
    int MP;
    int nslots;                // wave-uniform: slots any row of this wave uses (the code tests slot 3 only: SLOT_ON)
    int n8w;                   // wave-uniform: operation types of the wave's largest instance, rounded up to 8 (length of a walk)
    // lane = operation types 16 s + l.  Slots without an operation type carry kind 15 (no job lane: its word reads
    // 0xFFFFFFFF = "stage 255"): never unassigned, and their fluid numbers are all 0
    uint32_t kB[GS], em[GS];   // stage | J_r << 8 | kind << 16 | flags << 24;  elig | fmask << 8
    int due[GS];
    double rsum[GS], tsum[GS];
    uint32_t asgw;             // byte s: machine operation type 16 s + l was assigned to (0xFF: not yet)
    // lane = job: its state word (0xFFFFFFFF beyond njobs), first operation type | J_r << 8, due date, and of the
    // job's current operation type (first + next stage): elig | fmask << 8, fluid_time_sum, fluid_rate_sum
    uint32_t jwl, jinfo, emc;
    int duej;
    double tsumc, rsumc;
    // lane = machine
    int tend, mjob, mcnt;      // mcnt: operation types the machine can process (static)
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
    uint32_t res;              // fused kernel, small batches: byte offset (from g_lds) of this row's RESIDENT copy of the instance's
                               // fluid numbers, {arrival, rate} table and processing times (g_make_resident)
    int resident;              // wave-uniform (a kernel argument): the rows have resident copies; 0 = read from memory
};

// the (operation x machine) rows {arrival, rate} of a lane's slots, in flight while the step decides (g_cols_issue)
template <int MPC>
struct GCols { double2 c[GS][MPC]; };

// the current operation type of every job lane: k = first + next stage, and its static data out of the operation slots
template <int V, bool EARLY>
GDEV void g_gather_current(GE<V> &e, const DevBatch &b, bool want_rsum) {
    const int kc = (int)(e.jinfo & 0xFFu) + (int)(e.jwl & 0xFFu);
    const int src = kc & 15, sl = (kc >> 4) & 3;
    if (!EARLY) {
        // large batches: the fluid numbers of the operation slots are not kept in registers between the start of a step and its
        // observation (resident waves hide the second fetch; the registers would cost a wave per SIMD): every job lane fetches
        // {fluid_rate_sum, fluid_time_sum} of its current operation type, the elig | fmask word comes out of the slots
        double2 rt;
        if (e.resident) rt = res_B(e.res)[16 * sl + src];
        else rt = reinterpret_cast<const double2 *>(e.ir + b.L.i_op + sl * 512 + 256)[src];
        uint32_t a[GS];
#pragma unroll
        for (int s = 0; s < GS; ++s) { a[s] = 0; if (SLOT_ON(e, s)) a[s] = greadu(e.em[s], src, e.gb); }
        e.emc = pick_slot(a, sl);
        e.rsumc = rt.x; e.tsumc = rt.y;
        return;
    }
    uint32_t a[GS], tl[GS], th[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        a[s] = 0; tl[s] = 0; th[s] = 0;
        if (SLOT_ON(e, s)) {
            a[s] = greadu(e.em[s], src, e.gb);
            tl[s] = greadu((uint32_t)__double2loint(e.tsum[s]), src, e.gb);
            th[s] = greadu((uint32_t)__double2hiint(e.tsum[s]), src, e.gb);
        }
    }
    e.emc = pick_slot(a, sl);
    e.tsumc = __hiloint2double((int)pick_slot(th, sl), (int)pick_slot(tl, sl));
    e.rsumc = 0.0;
    if (want_rsum) {
        uint32_t rl[GS], rh[GS];
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            rl[s] = 0; rh[s] = 0;
            if (SLOT_ON(e, s)) {
                rl[s] = greadu((uint32_t)__double2loint(e.rsum[s]), src, e.gb);
                rh[s] = greadu((uint32_t)__double2hiint(e.rsum[s]), src, e.gb);
            }
        }
        e.rsumc = __hiloint2double((int)pick_slot(rh, sl), (int)pick_slot(rl, sl));
    }
}

// random.choice replacement (fjsp_oracle.h): index into a list of length n; `take` rows consume one draw
template <int V>
GDEV int g_rng(GE<V> &e, bool take, int n) {
    const uint64_t u = splitmix64(e.env_seed + (uint64_t)e.rng_calls);
    e.rng_calls += take ? 1u : 0u;
    return (int)__umulhi((uint32_t)(u >> 32), (uint32_t)n);
}

// SO_FJSSP.py:267-298 task_select for every row with go, on the job lanes: a job's current operation type is available
// when the job waits (it carries a FIFO sequence) and an eligible machine is idle; the lists the reference builds in
// kind_task_tuple order are, restricted to those, in job order.  Every rule is "the first candidate of the largest
// key".  Returns the chosen JOB lane or -1 (status set).
template <int V>
GDEV int g_task_select(GE<V> &e, bool go, int a0, uint32_t idle) {
    const bool waiting = e.jwl < kSeqNone;
    const bool av = waiting && (e.emc & idle) != 0, fav = waiting && (e.emc & (idle << 8)) != 0;
    const uint32_t avm = gballot(av, e.gb), favm = gballot(fav, e.gb);
    e.status |= (go && avm == 0) ? (uint32_t)FJSP_ST_NO_EVENT : 0u;
    e.status |= (go && avm != 0 && a0 >= 6) ? (uint32_t)FJSP_ST_BAD_TASK_RULE : 0u;          // MyError :297
    const bool run = go && avm != 0 && a0 < 6;
    // update_parameter for a list of one (:126-154): estimated delay = max_e = urgency, actual delay = max_a
    const double de = ((double)e.t + e.tsumc) - (double)e.duej;          // :136,139 with task_index 0
    const int da = e.t - e.duej;                                         // :138
    const bool pre = av && (a0 == 0 ? de > 0.0 : (a0 == 1 && da > 0));   // rules 1, 2: the delayed types first (:269-278)
    const uint32_t prem = gballot(pre, e.gb);
    bool C = av;
    C = (a0 <= 1 && prem != 0) ? pre : C;
    C = (a0 >= 2 && a0 <= 4 && favm != 0) ? fav : C;                     // rules 3-5: the fluid-available types first (:279-293)
    uint64_t key = sortable(de);
    if (wave_any(run && a0 == 2)) {
        const double fq = 1.0 - e.rsumc * (double)e.t;                   // fluid_unprocessed_number (:239-240), Q0 = 1 job
        key = a0 == 2 ? sortable(1.0 - fq) : key;                        // Tasks.gap class_FJSSP.py:70-72 (one unprocessed task)
    }
    key = (a0 == 1 && prem != 0) ? sortable_max_i32(da) : key;           // :274-278
    key = a0 == 4 ? sortable_min_i32(e.duej) : key;                      // :289-293
    key = a0 == 5 ? 0ull : key;
    int r = first_max(C && run, key, e.gb);
    if (wave_any(run && a0 == 5)) {
        // random.choice(task_available_list) (:294-295): the idx-th available job
        const bool rnd = run && a0 == 5;
        const int idx = g_rng(e, rnd, __builtin_popcount(avm));
        const uint32_t pick = gballot(av && __builtin_popcount(avm & ((1u << e.l) - 1u)) == idx, e.gb);
        r = rnd ? (int)__builtin_ctz(pick | 0x10000u) : r;
    }
    return (run && r < 16) ? r : -1;
}

// Strictly sequential float sum of n8 (a multiple of 8, >= 8, wave-uniform) operands of an LDS row, left to right like the
// reference's sum(): the ring of fjsp_common.h's lds_chain_sum_ring8 (eight 16-byte registers, each refilled 16 elements
// ahead) with ONE counted wait per two registers -- LDS reads return in order, so lgkmcnt(6) with eight reads in flight says
// the two oldest have landed -- which takes the walk from 2.8 to 1.9 instructions per element.  Reads up to 16 entries past n8
// (never consumed): rows are followed by at least 128 bytes of the same allocation.
#define FJSP_LDS_WAIT2(r0, r1, cnt) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(cnt))
GDEV double row_chain_sum(const double *src, int n8) {
    double acc = 0.0;
    uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(src);      // the low half of a flat LDS address is the LDS offset
    const int n = __builtin_amdgcn_readfirstlane(n8);
    fjsp_d2 A0, A1, A2, A3, A4, A5, A6, A7;
    FJSP_LDS_READ128(A0, a, 0); FJSP_LDS_READ128(A1, a, 16); FJSP_LDS_READ128(A2, a, 32); FJSP_LDS_READ128(A3, a, 48);
    FJSP_LDS_READ128(A4, a, 64); FJSP_LDS_READ128(A5, a, 80); FJSP_LDS_READ128(A6, a, 96); FJSP_LDS_READ128(A7, a, 112);
    int i = 0;
#define FJSP_RING_PAIR(R0, R1, off)                                                        \
    FJSP_LDS_WAIT2(R0, R1, 6);                                                             \
    acc = acc + R0.x; acc = acc + R0.y; acc = acc + R1.x; acc = acc + R1.y;                \
    FJSP_LDS_READ128(R0, a, off); FJSP_LDS_READ128(R1, a, off + 16);
    for (; i + 16 <= n; i += 16) {
        FJSP_RING_PAIR(A0, A1, 128) FJSP_RING_PAIR(A2, A3, 160) FJSP_RING_PAIR(A4, A5, 192) FJSP_RING_PAIR(A6, A7, 224)
        a += 128;
    }
#undef FJSP_RING_PAIR
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7));
    if (i < n) {                         // n is a multiple of 8: eight operands left, in the first half of the ring
        acc = acc + A0.x; acc = acc + A0.y; acc = acc + A1.x; acc = acc + A1.y;
        acc = acc + A2.x; acc = acc + A2.y; acc = acc + A3.x; acc = acc + A3.y;
    }
    return acc;
}

// Strictly sequential sums out of the LDS rows: lanes with `walk` sum n8w entries of their row (rows are zero padded to
// n8w: +0.0 is an exact identity of a running sum that starts at +0.0), all of them side by side.
template <int V>
GDEV double g_walk(const GE<V> &e, bool walk, int row) {
    double sm = 0.0;
    if (wave_any(walk)) sm = row_chain_sum(e.rows + (walk ? row : 0) * KS, e.n8w);
    return sm;
}

// Request the {arrival, rate} entries of every slot of the rows with `need`, for the IDLE machines (every candidate list of
// machine_select is a subset of them), from the machine-major copy of the table (Layout::i_colm): class_FJSSP.py:144-146 reads a
// whole machine at a time
template <int V, int MPC>
GDEV void g_cols_issue(const GE<V> &e, const DevBatch &b, bool need, GCols<MPC> &cr) {
    const double2 *colm = reinterpret_cast<const double2 *>(e.ir + b.L.i_colm);
    const uint32_t idle = need ? (~e.busy & e.mmask) : 0u;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
#pragma unroll
        for (int m = 0; m < MPC; ++m) {
            cr.c[s][m] = make_double2(0.0, 0.0);
            if (SLOT_ON(e, s) && ((idle >> m) & 1u)) cr.c[s][m] = colm[m * 64 + 16 * s + e.l];
        }
    }
}
// Machine.gap_ave's operands (class_FJSSP.py:137-146): lane (s, l) writes unprocessed - fluid_unprocessed of its
// operation type on machine m to LDS row m.  The (k, m) entries of types that cannot run on m hold arrival = rate = 0
// (fluid_tables_kernel), so their gap is 0.0 - (0.0 - dt * 0.0) = +0.0 -- an exact identity of the running sum, like
// the padding beyond K -- and no eligibility test is needed.
template <int V, int MPC>
GDEV void g_gap_rows(const GE<V> &e, bool need, const GCols<MPC> &cr) {
    const double dt = (double)e.t;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (SLOT_ON(e, s) && need) {
            const uint32_t asg = (e.asgw >> (8 * s)) & 0xFFu;
            double *dst = e.rows + 16 * s + e.l;
#pragma unroll
            for (int m = 0; m < MPC; ++m) {
                const double2 ar = cr.c[s][m];
                // (unprocessed = arrival, less one where this type was assigned to m: class_FJSSP.py:198, :304)
                const double un = asg == (uint32_t)m ? ar.x - 1.0 : ar.x;
                dst[m * KS] = un - (ar.x - dt * ar.y);
            }
        }
    }
}

// Machine.gap_ave (class_FJSSP.py:144-146) of the machines in C for the rows with `need`, large-batch form: the registers
// of GCols and five LDS rows per environment would cost resident waves, and it is the resident waves that hide the memory
// there.  So the {arrival, rate} entries are fetched here, for the CANDIDATE machines only, three machines (= three LDS
// rows per environment) per pass; lanes 0..2 of the row walk them and hand the sums to the machines' lanes.  Returns the
// machine lanes' gap_ave (lanes of machines in C).
template <int V>
GDEV double g_gap_ave_lean(const GE<V> &e, const DevBatch &b, bool need, uint32_t C) {
    const double dt = (double)e.t;
    const double2 *colm_g = reinterpret_cast<const double2 *>(e.ir + b.L.i_colm);        // machine-major: [m][64]
    const double2 *colm_l = res_C(e.res);
    auto colm_at = [&](int i) -> double2 { if (e.resident) return colm_l[i]; return colm_g[i]; };
    uint32_t rest = need ? C : 0u;
    double sum_m = 0.0;
    while (wave_any(rest != 0)) {
        // this pass's machines (row-uniform; 8 = none)
        uint32_t r1 = rest & (rest - 1), r2 = r1 & (r1 - 1);
        const int m0 = rest ? (int)__builtin_ctz(rest) : 8, m1 = r1 ? (int)__builtin_ctz(r1) : 8, m2 = r2 ? (int)__builtin_ctz(r2) : 8;
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            if (SLOT_ON(e, s) && rest != 0) {
                const int asg = (int)((e.asgw >> (8 * s)) & 0xFFu);
                const int k = 16 * s + e.l;
                double *dst = e.rows + k;
                // (entries of types that cannot run on the machine hold arrival = rate = 0: their gap is +0.0, see g_gap_rows)
                const double2 a0 = colm_at((m0 & 7) * 64 + k);
                const double2 a1 = m1 < 8 ? colm_at(m1 * 64 + k) : make_double2(0.0, 0.0), a2 = m2 < 8 ? colm_at(m2 * 64 + k) : make_double2(0.0, 0.0);
                dst[0] = (asg == m0 ? a0.x - 1.0 : a0.x) - (a0.x - dt * a0.y);          // class_FJSSP.py:198, :304, :137-142
                dst[KS] = (asg == m1 ? a1.x - 1.0 : a1.x) - (a1.x - dt * a1.y);
                dst[2 * KS] = (asg == m2 ? a2.x - 1.0 : a2.x) - (a2.x - dt * a2.y);
            }
        }
        lds_sync();
        const int mine = e.l == 0 ? m0 : (e.l == 1 ? m1 : (e.l == 2 ? m2 : 8));
        const double sm = g_walk<V>(e, rest != 0 && mine < 8, e.l);
        const double s0 = bcd<0>(sm), s1 = bcd<1>(sm), s2 = bcd<2>(sm);
        sum_m = e.l == m0 ? s0 : (e.l == m1 ? s1 : (e.l == m2 ? s2 : sum_m));
        lds_sync();
        rest = r2 & (r2 - 1);
    }
    return sum_m / ((double)e.mcnt + 1e-18);
}

// SO_FJSSP.py:300-322 / MO_FJSSP_discretes.py:209-230 machine_select for the rows with `go`.  Machine ids < 8: every
// CPython set involved iterates in ascending order (fjsp_pyset.h), the candidate lists are bit masks and "first
// extremum wins" is the lowest machine lane that attains it.  gap_rows: the rows whose LDS rows hold Machine.gap_ave's
// operands.  Returns m or -1 (status set); *pm_out its processing time.
template <int V, bool EARLY>
GDEV int g_machine_select(GE<V> &e, const DevBatch &b, bool go, int a1, int k_sel, uint32_t em_sel, uint32_t idle, bool gap_rows,
                          int *pm_out) {
    constexpr bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES;
    const uint32_t sel = idle & em_sel, fsel = idle & (em_sel >> 8);         // (idle has the low 8 bits only)
    e.status |= (go && sel == 0) ? (uint32_t)FJSP_ST_NO_EVENT : 0u;
    const bool bad = go && sel != 0 && a1 >= (is_mo ? 3 : 5);
    e.status |= bad ? (uint32_t)FJSP_ST_BAD_MACHINE_RULE : 0u;                // MyError :321
    const bool run = go && sel != 0 && !bad;
    const uint32_t fl = fsel ? fsel : sel;
    // per row: candidate list C and the kind of key (largest gap / least time / largest gap_ave / random)
    bool m_pt, m_gave, m_rnd;
    uint32_t C;
    if (is_mo) {                                                              // MO_FJSSP_discretes.py:213-227
        m_pt = a1 == 0 && fsel == 0; m_gave = a1 == 1; m_rnd = false;
        C = fl;
    } else {                                                                  // SO_FJSSP.py:304-319
        m_pt = a1 == 2; m_gave = a1 == 3; m_rnd = a1 == 4;
        C = (a1 == 0 || a1 == 3) ? fl : sel;
    }
    C = run ? C : 0u;
    const bool mem = ((C >> e.l) & 1u) != 0;            // (C has bits below 8 only)
    int pm = 0;
    double g = 0.0;
    if (((run ? sel : 0u) >> e.l) & 1u) {
        // lane m: p[m][k_sel] and gap_rj_dict[m][k_sel] (class_FJSSP.py:137-142); op-major layout: the column of k_sel is MP
        // contiguous entries per array.  k_sel has not been dispatched yet -- it is available -- so its unprocessed is
        // its arrival
        const int o = k_sel * e.MP + e.l;
        double2 ar;
        if (e.resident) { pm = res_P(e.res)[o]; ar = res_C(e.res)[e.l * 64 + k_sel]; }   // (the machine-major copy holds the same entries)
        else { pm = reinterpret_cast<const uint16_t *>(e.ir + b.L.i_p)[o]; ar = reinterpret_cast<const double2 *>(e.ir + b.L.i_col)[o]; }
        g = ar.x - (ar.x - (double)e.t * ar.y);
    }
    const bool need3 = m_gave && gap_rows && (C & (C - 1)) != 0;     // (a list of one is returned without ranking it)
    if (wave_any(need3)) {
        // Machine.gap_ave (class_FJSSP.py:144-146): the strictly sequential sum of the machine's row / (n + 1e-18)
        double gave;
        if (EARLY) gave = g_walk<V>(e, need3 && e.l < e.M, e.l) / ((double)e.mcnt + 1e-18);       // (rows: g_gap_rows)
        else gave = g_gap_ave_lean<V>(e, b, need3, C);
        g = need3 ? gave : g;
    }
    uint64_t key = sortable(g);
    key = m_pt ? sortable_min_i32(pm) : key;
    key = (m_rnd || (m_gave && !need3)) ? 0ull : key;   // no ranking: every member ties
    int m = first_max(mem, key, e.gb);
    if (wave_any(run && m_rnd)) {
        const bool rnd = run && m_rnd;
        const int idx = g_rng(e, rnd, __builtin_popcount(sel));                                                   // :318-319
        const uint32_t pick = gballot(mem && __builtin_popcount(C & ((1u << e.l) - 1u)) == idx, e.gb);
        m = rnd ? (int)__builtin_ctz(pick | 0x10000u) : m;
    }
    const bool ok = run && m < 16;
    *pm_out = gread(pm, ok ? m : 0, e.gb);
    return ok ? m : -1;
}

// SO_FJSSP.py:176-250: dispatch job r_sel (its current operation type k_sel) on m_sel, then advance the clock until some
// operation type is available again (or the episode ends), for the rows with `go`.  The event loop runs on the job and
// machine lanes alone.
template <int V>
GDEV void g_dispatch_advance(GE<V> &e, bool go, int r_sel, int k_sel, int m_sel, int pm) {
    // the job's word, first operation | J_r, due date: from its lane
    const int rs = go ? r_sel : 0;
    const uint32_t jw_sel = greadu(e.jwl, rs, e.gb), ji_sel = greadu(e.jinfo, rs, e.gb);
    const int due_sel = gread(e.duej, rs, e.gb);
    // elig | fmask of the job's NEXT operation type (k_sel + 1): what the event loop tests once the job waits again
    const int kn = (k_sel + 1) & 63;
    const uint32_t em_next = greadu(pick_slot(e.em, kn >> 4), kn & 15, e.gb);
    const int nj = (int)(jw_sel & 0xFFu) + 1, Jr = (int)((ji_sel >> 8) & 0xFFu);   // the job is at stage nj - 1; it moves on
    const bool last = nj == Jr;
    const int time_end = e.t + pm;                                           // :184
    const bool jmine = go && e.l == r_sel;
    e.jwl = jmine ? jst_pack(kNoSeq, (uint32_t)nj) : e.jwl;                  // :186-191
    e.emc = jmine ? em_next : e.emc;
    const int ksel_go = go ? k_sel : -1;
#pragma unroll
    for (int s = 0; s < GS; ++s)      // :198 unprocessed_rj_dict[m][(r, j)] -= 1, kept as "the machine this type was assigned to"
        e.asgw = ksel_go == 16 * s + e.l ? ((e.asgw & ~(0xFFu << (8 * s))) | ((uint32_t)m_sel << (8 * s))) : e.asgw;
    // machine lane: time_end, and job | (k of the job's next stage + 1) << 16 (0 in the high half: no further stage)
    const bool mine = go && e.l == m_sel;
    e.tend = mine ? time_end : e.tend;                                       // :194-197
    e.mjob = mine ? (r_sel | ((last ? 0 : k_sel + 2) << 16)) : e.mjob;
    e.busy |= go ? 1u << (m_sel & 31) : 0u;
    e.completion = go ? max(e.completion, time_end) : e.completion;
    {                                                                        // :200-202
        const int late = time_end - due_sel;
        e.n_un -= (go && last) ? 1 : 0;
        e.tard_done += (go && last && late > 0) ? late : 0;
    }
    bool act = go;
    for (;;) {
        const uint32_t idle = ~e.busy & e.mmask;
        act = act && gballot(e.jwl < kSeqNone && (e.emc & idle) != 0, e.gb) == 0;   // :204
        if (!wave_any(act)) break;
        const int tn = gmin((e.l < e.M && e.tend > e.t) ? e.tend : 0x7fffffff);   // :205-207 next event
        const bool noev = act && tn == 0x7fffffff;
        e.status |= noev ? (uint32_t)FJSP_ST_NO_EVENT : 0u;
        act = act && !noev;
        e.t = act ? tn : e.t;
        // :209-215 completions in ascending machine order: the job goes to the FIFO of its next stage (if any)
        const bool rel = act && e.l < e.M && e.tend == tn && (e.mjob >> 16) != 0;
        const uint32_t relm = gballot(rel, e.gb);
        const int rank = __builtin_popcount(relm & ((1u << e.l) - 1u));
        // every releasing machine lane hands its job lane the job's place in the append order (ds_permute: a push; the
        // other lanes push a zero to lane 15, which is no job lane -- at most 15 jobs)
        const int got = __builtin_amdgcn_ds_permute((e.gb + (rel ? (e.mjob & 0xFFFF) : 15)) << 2, rel ? rank + 1 : 0);
        e.jwl = (got != 0 && e.l < 15) ? ((e.jwl & 0xFFu) | ((e.seq_ctr + (uint32_t)got - 1u) << 8)) : e.jwl;
        e.seq_ctr += (uint32_t)__builtin_popcount(relm);
        const uint32_t freed = gballot(e.l < e.M && e.tend <= e.t, e.gb);   // :233-235
        e.busy &= act ? ~freed : 0xFFFFFFFFu;
        const bool fin = act && e.n_un == 0;                                 // :247-250
        e.done = fin ? 1 : e.done;
        act = act && !fin;
    }
}

// SO_FJSSP.py:78-97 state_extract and update_parameter's statistics (:99-166) at the current clock.  Returns the
// observation in the observation lanes (lane (8 + i) & 15 holds entry i) and delay_time_sum_unprocessed (:110-122)
// through *tard_unproc; only the rows with `on` walk their sums (the others' results are garbage nobody reads).
// stats_only: a step that hands no state back needs the tardiness for its reward and nothing else.
template <int V, bool EARLY>
GDEV double g_observe(const GE<V> &e, const DevBatch &b, bool on, bool stats_only, long long *tard_unproc) {
    using P = ObsPos<V>;
    constexpr int L_ave0 = (P::ave0 + 8) & 15, L_ave1 = (P::ave1 + 8) & 15, L_ave2 = (P::ave2 + 8) & 15;
    constexpr int L_sd0 = (P::sd0 + 8) & 15, L_sd1 = (P::sd1 + 8) & 15, L_sd2 = (P::sd2 + 8) & 15;
    // ---- per job (lists of one, :126-154): operations left, lateness; packed row sums
    //   c1 = tasks | delay_a << 8 | job_a << 16 | delay_e << 24      c2 = tardiness, low 16 bits      c3 = high bits | job_e << 24
    const uint32_t Jr = (e.jinfo >> 8) & 0xFFu, njl = e.jwl & 0xFFu;
    const uint32_t rem = njl < Jr ? Jr - njl : 0u;                  // (lanes beyond njobs: J_r = 0)
    const bool latej = e.t > e.duej;                                // :134-135 (:120-122 for the last stage)
    const uint32_t tl = (latej && rem != 0) ? (uint32_t)(e.t - e.duej) : 0u;
    uint32_t c1 = rem + (latej ? rem << 8 : 0u) + ((latej && rem != 0) ? 1u << 16 : 0u);
    uint32_t c2 = tl & 0xFFFFu, c3 = tl >> 16;
    *tard_unproc = 0;
    if (stats_only) {
        c2 = (uint32_t)gsum((int)c2); c3 = (uint32_t)gsum((int)c3);
        *tard_unproc = (long long)c2 + ((long long)c3 << 16);
        return 0.0;
    }
    // ---- per operation type: is it unassigned (stage of its job <= its own), its estimated delay (:136-137), its
    // gap_rate row entry; finish_rate is 0 or 1 here and the machines' time_end are integers: their left-to-right f64 sums
    // are exact, i.e. equal to the integer sums; gap_rate's sum is walked
    const double td = (double)e.t;
    bool in[GS];
    double grv[GS];
    uint32_t q0h[GS];
    double2 rt[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        rt[s] = make_double2(e.rsum[s], e.tsum[s]);
        if (!EARLY) {       // (large batches: the fluid numbers are fetched again, see g_gather_current)
            rt[s] = make_double2(0.0, 0.0);
            if (SLOT_ON(e, s) && 16 * s + e.l < e.K) {
                if (e.resident) rt[s] = res_B(e.res)[16 * s + e.l];
                else rt[s] = reinterpret_cast<const double2 *>(e.ir + b.L.i_op + s * 512 + 256)[e.l];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        in[s] = false; grv[s] = 0.0; q0h[s] = 0;
        if (SLOT_ON(e, s)) {
            const uint32_t kb = e.kB[s];
            const int kind = (int)((kb >> 16) & 0xFu);
            const uint32_t njk = greadu(e.jwl, kind, e.gb) & 0xFFu;
            const int due_s = gread(e.duej, kind, e.gb);                          // (one job per kind: the type's due date is its job's)
            in[s] = njk <= (kb & 0xFFu);
            const double de = (td + rt[s].y) - (double)due_s;
            const bool cnte = in[s] && de > 0.0;
            c1 += cnte ? 1u << 24 : 0u;
            c3 += (cnte && ((kb >> 24) & 1u)) ? 1u << 24 : 0u;
            q0h[s] = (kb >> 24) & 2u ? 0x3FF00000u : 0u;                                // Q0 = jobs of the kind: 1.0, or 0.0 for padding
            const double q0 = __hiloint2double((int)q0h[s], 0);
            const double fq = q0 - rt[s].x * td;                                        // fluid_unprocessed_number (:239-240)
            grv[s] = (in[s] ? 1.0 : 0.0) - fq;                                          // gap_rate class_FJSSP.py:66-68 (x / 1.0 == x)
            e.rows[16 * s + e.l] = grv[s];
        }
    }
    c1 = (uint32_t)gsum((int)c1); c2 = (uint32_t)gsum((int)c2); c3 = (uint32_t)gsum((int)c3);
    const int tasks = (int)(c1 & 0xFFu), delay_a = (int)((c1 >> 8) & 0xFFu), job_a = (int)((c1 >> 16) & 0xFFu), delay_e = (int)(c1 >> 24);
    const int job_e = (int)(c3 >> 24), jobs = e.n_un;
    *tard_unproc = (long long)c2 + ((long long)(c3 & 0xFFFFFFu) << 16);
    const uint32_t tlo = (uint32_t)gsum((e.l < e.M) ? (e.tend & 0xFFFF) : 0), thi = (uint32_t)gsum((e.l < e.M) ? (int)((uint32_t)e.tend >> 16) : 0);
    const double tsum_td = (double)(((long long)thi << 16) + (long long)tlo);
    lds_sync();
    const double cs1 = g_walk<V>(e, on && e.l == L_ave1, 0);
    lds_sync();
    const int oi = (e.l - 8) & 15, q = oi - P::ratio0;                            // q = 0..3: the ratios of :156-165
    const bool is_ratio = q >= 0 && q < 4;
    int inum = e.K - tasks, iden = e.K;                                           // finish_rate: types with their one task assigned
    inum = is_ratio ? (q == 0 ? delay_a : (q == 1 ? delay_e : (q == 2 ? job_a : job_e))) : inum;
    iden = is_ratio ? (q < 2 ? tasks : jobs) : iden;
    iden = e.l == L_ave2 ? e.M : iden;
    double num = (double)inum;
    num = e.l == L_ave1 ? cs1 : num;
    num = e.l == L_ave2 ? tsum_td : num;
    const double q1 = num / (double)iden;
    const double ave_fr = bcd<L_ave0>(q1), ave_gr = bcd<L_ave1>(q1), ave_td = bcd<L_ave2>(q1);
    // ---- second pass: squared deviations (math.pow(d, 2), :86-95), population standard deviations; their sums by the fixed tree
    // of fjsp_common.h (row_tree_sum_f64: the one sum that is not taken in the reference's order)
    const double f0 = (0.0 - ave_fr) * (0.0 - ave_fr), f1 = (1.0 - ave_fr) * (1.0 - ave_fr);
    double p_fr = 0.0, p_gr = 0.0;
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        if (SLOT_ON(e, s)) {
            const double q0 = __hiloint2double((int)q0h[s], 0);          // 1.0, or 0.0 for padding (x * 0.0 = 0.0: an exact identity)
            const double d2 = grv[s] - ave_gr;
            p_fr = p_fr + (in[s] ? f0 : f1) * q0;
            p_gr = p_gr + (d2 * d2) * q0;
        }
    }
    double d3 = (double)e.tend - ave_td;
    d3 = e.l < e.M ? d3 * d3 : 0.0;
    const double cs_fr = row_tree_sum_f64(p_fr), cs_gr = row_tree_sum_f64(p_gr), cs_td = row_tree_sum_f64(d3);
    const double num2 = e.l == L_sd0 ? cs_fr : (e.l == L_sd1 ? cs_gr : cs_td);
    const double q2 = sqrt(num2 / (double)(e.l == L_sd2 ? e.M : e.K));
    double cur = (is_ratio && e.done) ? 0.0 : q1;                                 // the ratios are 0 once the episode is over
    cur = (e.l == L_sd0 || e.l == L_sd1 || e.l == L_sd2) ? q2 : cur;
    if (is_so_v<V>) cur = oi == 0 ? (double)e.M : cur;
    return cur;
}

// v(t-1) of the state vector is the observation of the environment as a step finds it; steps that handed no state back did
// not keep it current (fjsp_kernels.hip obs_refresh).  (Inline like everything else: a call would pin the row state in scratch.)
template <int V, bool EARLY>
GDEV void g_obs_refresh(GE<V> &e, const DevBatch &b, bool on) {
    long long tu;
    const double v = g_observe<V, EARLY>(e, b, on, false, &tu);
    if (on && ((e.l - 8) & 15) < kNObs<V>) e.obs_prev = v;
    if (on) e.misc &= ~(1u << 24);
}

// Reward and bookkeeping of step() (SO_FJSSP.py:259-265, MO_FJSSP_discretes.py:232-244) once the observation is out.
template <int V>
GDEV double g_reward(GE<V> &e, bool go, const MoW &mo, long long tard_unproc) {
    const long long delay_new = e.tard_done + tard_unproc;                       // :259
    const long long delta = delay_new - e.delay_sum;
    const int dc = e.completion_last - e.completion;
    e.delay_sum = go ? delay_new : e.delay_sum;                                  // :263
    e.completion_last = go ? e.completion : e.completion_last;
    double r;
    if (V != FJSP_VARIANT_MO_FJSSP_DISCRETES) r = (double)(-delta);              // :328 (exact integer)
    else {
        const double w0 = mo.w0, w1 = mo.w1, cn = mo.cn, tn = mo.tn;
        if (cn > 0.0 && tn > 0.0) r = (double)dc / cn * w0 + (double)(-delta) / tn * w1;
        else if (w1 == 1.0) r = (double)(-delta);
        else if (w0 == 1.0) r = (double)dc;
        else { r = 0.0; e.status |= go ? (uint32_t)FJSP_ST_BAD_TASK_RULE : 0u; }   // MyError :244
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
// others (bounds come from the kernel arguments), one memory round trip.  `between` runs after the loads are out and
// before they are waited for (the kernels request the gap_ave rows there).
template <int V, bool EARLY, class F>
GDEV void g_open(GE<V> &e, const DevBatch &b, int wave_id, unsigned char *lds, int rows_per_env, F &&between) {
    using FO = FixedOffsets;
    const int lane = (int)__lane_id();
    e.l = lane & 15; e.gb = lane & 48;
    const int env_raw = wave_id * 4 + (lane >> 4);
    e.live = env_raw < b.N;
    e.env = min(env_raw, b.N - 1);
    const int MP = b.MP, JP = b.JP;
    e.MP = MP;
    int inst = e.env;
    if (b.n_inst != b.N) {                  // (a real branch: the ~25-instruction integer modulo stays off the usual path)
        asm volatile("" ::: "memory");
        inst = e.env % b.n_inst;
    }
    const unsigned char *ir = b.inst + (size_t)inst * b.L.i_stride;
    unsigned char *er = b.envs + (size_t)e.env * FO::e_stride_plain((uint32_t)MP, (uint32_t)JP, 64u, true);
    e.ir = ir; e.er = er;
    e.rows = reinterpret_cast<double *>(lds) + (size_t)(lane >> 4) * rows_per_env * KS;
    e.res = 0u; e.resident = 0;
    // ---- issue every load.  Operation rows: up to the batch's largest instance (small batches: one round trip), or up to
    // THIS environment's operation count, fetched first (large batches: a dependent fetch of one byte buys 2-4 lines of padding)
    int kq = b.kmax;
    if (!EARLY && !b.kenv_first) kq = min(b.kmax, 48);       // (the fourth slot, if this environment has one: below, once K is known)
    if (!EARLY && b.kenv_first) kq = (int)b.kenv[e.env];         // (the count first, then only the words it has; A/B knob FJSP_GROUP_KENV=0)
    const unsigned char *op = ir + b.L.i_op;
    const uint32_t hw = reinterpret_cast<const uint32_t *>(op + 2048)[e.l];        // machine / job / instance words
    const int duej = reinterpret_cast<const int32_t *>(op + 2048 + 64)[e.l];
    uint4 A[GS];
    double2 B[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        A[s] = make_uint4(0u, 0u, 0u, 0u); B[s] = make_double2(0.0, 0.0);
        if (16 * s + e.l < kq) {
            const unsigned char *slot = op + s * 512;
            // the two operation words from the 8-byte copy (the due date comes from the job's lane, g_observe)
            const uint2 a8 = reinterpret_cast<const uint2 *>(ir + b.L.i_op8)[16 * s + e.l];
            A[s] = make_uint4(a8.x, a8.y, 0u, 0u);
            if (EARLY) B[s] = reinterpret_cast<const double2 *>(slot + 256)[e.l];      // (large batches: g_gather_current, g_observe)
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
    // (the slots in use are known from the batch until the instance words arrive: the requests of `between` cover them all)
    e.nslots = (b.kmax + 15) >> 4;
    between();
    // ---- consume
    e.K = (int)(bcu<0>(hw) >> 24); e.M = (int)(bcu<1>(hw) >> 24); e.njobs = (int)(bcu<2>(hw) >> 24);
    e.mmask = (1u << e.M) - 1u;
    {
        const int k0 = __builtin_amdgcn_readlane(e.K, 0), k1 = __builtin_amdgcn_readlane(e.K, 16);
        const int k2 = __builtin_amdgcn_readlane(e.K, 32), k3 = __builtin_amdgcn_readlane(e.K, 48);
        const int kw = max(max(k0, k1), max(k2, k3));
        e.nslots = (kw + 15) >> 4;
        e.n8w = max((kw + 7) & ~7, 8);
    }
    if (!EARLY && !b.kenv_first && b.kmax > 48 && e.nslots > 3) {
        // more than 48 operation types in one of the wave's environments (rare at 10x5): one more round trip for the fourth slot
        if (48 + e.l < e.K) { const uint2 a8 = reinterpret_cast<const uint2 *>(ir + b.L.i_op8)[48 + e.l]; A[3] = make_uint4(a8.x, a8.y, 0u, 0u); }
    }
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        e.kB[s] = (A[s].x >> 24) & 2u ? A[s].x : 0x000F0000u;      // (no operation type: kind 15, see GE)
        e.em[s] = A[s].y; e.due[s] = (int)A[s].z;
        e.rsum[s] = B[s].x; e.tsum[s] = B[s].y;
    }
    e.mcnt = (int)(hw & 0xFFu);
    e.jinfo = (hw >> 8) & 0xFFFFu;
    e.duej = duej;
    e.asgw = asgw; e.jwl = e.l < e.njobs ? jwl : 0xFFFFFFFFu;
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
}

// One step() of the rows with go_in.  need_obs: the caller wants the state vector.  cr: the gap_ave rows requested for
// the rows with gap_need (g_cols_issue).  Returns the reward; *k_out / *m_out the chosen pair (-1: none).
#ifdef FJSP_GSTAMPS
#define GSTAMP_PARAM , unsigned long long (&gst)[14], unsigned long long &gst_t0
#define GSTAMP_ARG , gst, gst_t0
#else
#define GSTAMP_PARAM
#define GSTAMP_ARG
#endif
template <int V, int MPC, bool EARLY>
GDEV double g_step(GE<V> &e, const DevBatch &b, bool go_in, int a0, int a1, const MoW &mo, bool need_obs, double *state_out,
                   bool gap_need, const GCols<MPC> &cr, int *k_out, int *m_out GSTAMP_PARAM) {
    constexpr bool is_mo = V == FJSP_VARIANT_MO_FJSSP_DISCRETES;
    bool go = go_in;
    *k_out = -1; *m_out = -1;
#if defined(FJSP_GABLATE) && FJSP_GABLATE == 1
    return 0.0;                                                                   // diagnostic: state in / state out only
#endif
    const bool stale = (e.misc >> 24) & 1u;
    if (need_obs && wave_any(go && stale)) g_obs_refresh<V, EARLY>(e, b, go && stale);
    if (is_mo) {                     // flat action -> self.actions[action] (MO_FJSSP_discretes.py:26,92)
        e.status |= (go && a0 >= 18) ? (uint32_t)FJSP_ST_BAD_TASK_RULE : 0u;     // IndexError
        go = go && a0 < 18;
        a1 = a0 % 3; a0 = a0 / 3;
    }
    g_gather_current<V, EARLY>(e, b, wave_any(go && a0 == 2));
    GSTAMP(2);
    const uint32_t idle = ~e.busy & e.mmask;
    const int r_sel = g_task_select<V>(e, go, a0, idle);
    GSTAMP(3);
    if (EARLY && wave_any(gap_need)) {         // (after task_select: the rows requested at the start of the step have had time to arrive)
        g_gap_rows<V, MPC>(e, gap_need, cr);
        lds_sync();
    }
#if defined(FJSP_GABLATE) && FJSP_GABLATE == 2
    e.rng_calls += (uint32_t)r_sel; return 0.0;                                  // diagnostic: stop after task_select
#endif
    go = go && r_sel >= 0;
    const int rs = go ? r_sel : 0;
    GSTAMP(4);
    // the job's current operation type and its elig | fmask: from the job's lane
    const int k_sel = gread((int)(e.jinfo & 0xFFu) + (int)(e.jwl & 0xFFu), rs, e.gb);
    const uint32_t em_sel = greadu(e.emc, rs, e.gb);
    int pm = 0;
    const int m_sel = g_machine_select<V, EARLY>(e, b, go, a1, k_sel, em_sel, idle, gap_need, &pm);
    *k_out = go ? k_sel : -1; *m_out = m_sel;
#if defined(FJSP_GABLATE) && FJSP_GABLATE == 3
    e.rng_calls += (uint32_t)(m_sel + pm); return 0.0;                           // diagnostic: stop after machine_select
#endif
    go = go && m_sel >= 0;
    GSTAMP(5);
    g_dispatch_advance<V>(e, go, rs, k_sel, m_sel, pm);
    GSTAMP(6);
#if defined(FJSP_GABLATE) && FJSP_GABLATE == 4
    return 0.0;                                                                   // diagnostic: stop after dispatch_and_advance
#endif
    e.step_count += go ? 1 : 0;                                                  // :252
    long long tard_unproc = 0;
#if defined(FJSP_GABLATE) && FJSP_GABLATE == 5
    const double cur = g_observe<V, EARLY>(e, b, go, true, &tard_unproc);     // diagnostic: statistics only, no observation
#else
    const double cur = g_observe<V, EARLY>(e, b, go, !need_obs, &tard_unproc);   // :256
#endif
    GSTAMP(7);
    if (need_obs) g_emit<V>(e, go, cur, reinterpret_cast<const double *>(e.ir + b.L.i_ss), state_out);
    else e.misc |= go ? 1u << 24 : 0u;
    return g_reward<V>(e, go, mo, tard_unproc);
}

// does the rule pair of a row rank its candidate machines by Machine.gap_ave (SO_FJSSP.py:313-317 rule 4,
// MO_FJSSP_discretes.py:218-222 rule 2 of the flat action's pair)
template <int V>
GDEV bool rule_wants_gap_ave(uint32_t araw) {
    if (V == FJSP_VARIANT_MO_FJSSP_DISCRETES) return (araw & 0xFFu) < 18u && (araw & 0xFFu) % 3u == 1u;
    return (araw >> 8) == 3u;
}

// Copy the instance's static tables of this row's environment into its resident LDS block (once per fused launch)
template <int V, int MPC>
GDEV void g_make_resident(GE<V> &e, const DevBatch &b, uint32_t block_off) {
    unsigned char *dst = g_lds + block_off;
    double2 *B = reinterpret_cast<double2 *>(dst);
    uint4 *P = reinterpret_cast<uint4 *>(dst + 1024u);
    double2 *C = reinterpret_cast<double2 *>(dst + 2048u);
    const unsigned char *op = e.ir + b.L.i_op;
#pragma unroll
    for (int s = 0; s < GS; ++s) B[16 * s + e.l] = reinterpret_cast<const double2 *>(op + s * 512 + 256)[e.l];
    const uint4 *pg = reinterpret_cast<const uint4 *>(e.ir + b.L.i_p);                       // KP x MP u16, KP = 64: MP * 8 uint4
    for (int q = e.l; q < e.MP * 8; q += 16) P[q] = pg[q];
    const double2 *cg = reinterpret_cast<const double2 *>(e.ir + b.L.i_colm);
    for (int q = e.l; q < e.MP * 64; q += 16) C[q] = cg[q];
    lds_sync();
    e.res = block_off; e.resident = 1;
}

// ------------------------------------------------------------------------ kernels
// One step of every environment of a group batch (fjsp_kernels.hip step_kernel for the same batch gives the same results)
// EARLY: request the gap_ave rows together with the state (small batches: a wave is alone on its SIMD and nothing else hides
// the memory round trip); otherwise they are fetched where they are used and the kernel keeps to 128 registers
template <int V, int MPC, bool EARLY>
__global__ __launch_bounds__(EARLY ? 256 : 64, EARLY ? 1 : 4) void gstep_kernel(DevBatch b, const uint8_t *actions, const double *mo, int autoreset, double *state_out,
                                                   double *reward_out, uint8_t *done_out, int16_t *trace_km) {
    GE<V> e;
    GSTAMP_DECL;
    GSTAMP_BEGIN();
    // (small batches are launched four waves to a workgroup -- a quarter of the workgroups to dispatch; the waves never meet)
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_id = (int)blockIdx.x * (int)(blockDim.x >> 6) + wib;
    unsigned char *const my_lds = g_lds + (size_t)wib * group_lds_bytes<MPC, EARLY>();
    // the action pair of the row's environment (2-byte aligned: checked by the host entry points)
    const int env_raw = wave_id * 4 + (int)(__lane_id() >> 4);
    const int env0 = min(env_raw, b.N - 1);
    const uint32_t araw = reinterpret_cast<const uint16_t *>(actions)[env0];
    MoW mw = {0.0, 1.0, 0.0, 0.0};
    if (V == FJSP_VARIANT_MO_FJSSP_DISCRETES && mo) {
        const double2 m01 = *reinterpret_cast<const double2 *>(mo + (size_t)env0 * 4), m23 = *reinterpret_cast<const double2 *>(mo + (size_t)env0 * 4 + 2);
        mw.w0 = m01.x; mw.w1 = m01.y; mw.cn = m23.x; mw.tn = m23.y;
    }
    GCols<MPC> cr;
    bool gap_need = false;
    g_open<V, EARLY>(e, b, wave_id, my_lds, group_rows<MPC, EARLY>(), []() {});
    gap_need = env_raw < b.N && rule_wants_gap_ave<V>(araw);
    const int a0 = (int)(araw & 0xFFu), a1 = (int)(araw >> 8);
    bool go = e.live;
    GSTAMP(0);
    if (wave_any(go && e.done != 0)) {
        const bool was_done = go && e.done != 0;
        if (autoreset == 1) g_restart<V>(e, b, was_done);
        else {
            if (was_done && autoreset == 0) e.status |= FJSP_ST_STEP_AFTER_DONE;      // 2: idle silently
            go = go && !was_done;
        }
    }
    int k_sel = -1, m_sel = -1;
    GSTAMP(1);
    // small batches: Machine.gap_ave's operands are requested now (the machines that are idle are known), used after task_select
    if (EARLY && wave_any(gap_need && go)) g_cols_issue<V, MPC>(e, b, gap_need && go, cr);
    const double reward = g_step<V, MPC, EARLY>(e, b, go, a0, a1, mw, state_out != nullptr, state_out, gap_need && go, cr, &k_sel, &m_sel GSTAMP_ARG);
    if (e.live && e.l == 0) {
        if (reward_out) reward_out[e.env] = reward;
        if (done_out) done_out[e.env] = (uint8_t)e.done;
        if (trace_km) { trace_km[(size_t)e.env * 2] = (int16_t)k_sel; trace_km[(size_t)e.env * 2 + 1] = (int16_t)m_sel; }
    }
    g_store<V>(e, b);
    GSTAMP(8);
    GSTAMP_FLUSH();
}

// T fused steps per launch with the actions given (rule sweeps): the environments live in registers for the whole
// launch.  Same outputs as fjsp_kernels.hip rollout_kernel.
template <int V, int MPC, bool EARLY>
__global__ __launch_bounds__(EARLY ? 256 : 64, EARLY ? 1 : 2) void grollout_kernel(DevBatch b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km,
                                                      double *reward_out, double *state_last, int resident) {
    GE<V> e;
    GSTAMP_DECL;
    GSTAMP_BEGIN();
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_id = (int)blockIdx.x * (int)(blockDim.x >> 6) + wib;
    unsigned char *const my_lds = g_lds + (size_t)wib * group_lds_bytes<MPC, EARLY>();
    g_open<V, EARLY>(e, b, wave_id, my_lds, group_rows<MPC, EARLY>(), []() {});
    // (small batches, one wave to a workgroup: the static tables next to the rows, see res_bytes)
    if (!EARLY && resident) g_make_resident<V, MPC>(e, b, (uint32_t)group_lds_bytes<MPC, EARLY>() + (uint32_t)(__lane_id() >> 4) * res_bytes<MPC>());
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
        const bool gap_need = go && rule_wants_gap_ave<V>(araw);
        GCols<MPC> cr;
        if (EARLY) g_cols_issue<V, MPC>(e, b, gap_need, cr);
        int k_sel = -1, m_sel = -1;
        const double reward = g_step<V, MPC, EARLY>(e, b, go, (int)(araw & 0xFFu), (int)(araw >> 8), mw, state_last != nullptr, state_last,
                                             gap_need, cr, &k_sel, &m_sel GSTAMP_ARG);
        if (e.live && e.l == 0) {
            if (trace_km) { trace_km[o * 2] = (int16_t)k_sel; trace_km[o * 2 + 1] = (int16_t)m_sel; }
            if (reward_out) reward_out[o] = reward;
        }
    }
    g_store<V>(e, b);
}

}  // namespace grp

#ifdef FJSP_GSTAMPS
extern "C" int fjsp_debug_read_gstamps(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(grp::fjsp_gstamp_acc), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(grp::fjsp_gstamp_acc), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// ------------------------------------------------------------------ host launchers
// Which build of the step.  Per-step launches of batches that leave a SIMD a single wave (4 environments per wave, 1 024 SIMDs:
// up to 4 096 environments, a little beyond) run the variant that requests everything at the top -- nothing else hides a
// memory round trip there; larger batches the register-lean one (measured crossover between 4 096 and 6 144 environments:
// 8.09 vs 8.43 us at 4 096, 9.29 vs 9.08 us at 6 144).  The fused kernel always runs the lean build: its environment stays
// in registers and LDS across steps, the early build's requests are redundant there and its registers cost the second
// wave (4 096 envs: 771 -> 905 M env-steps/s with states, 1.08 -> 1.40 G without; 8 192: 809 M -> 1.49 G).
// FJSP_GROUP_EARLY=0/1 overrides both (A/B runs).
static int group_early_forced() {
    static const int forced = [] { const char *v = getenv("FJSP_GROUP_EARLY"); return v ? atoi(v) : -1; }();
    return forced;
}
static bool group_early(const DevBatch &b) { return group_early_forced() >= 0 ? group_early_forced() != 0 : b.N <= 5120; }
static bool group_early_rollout(const DevBatch &) { return group_early_forced() >= 0 ? group_early_forced() != 0 : false; }
// DIAGNOSTIC knob (A/B runs of the occupancy a batch size needs): FJSP_GROUP_LDS_PAD=<bytes> of extra dynamic LDS per wave
static size_t group_lds_pad() {
    static const size_t pad = [] { const char *v = getenv("FJSP_GROUP_LDS_PAD"); return v ? (size_t)atol(v) : (size_t)0; }();
    return pad;
}
template <typename K>
static inline void group_allow_lds(K kernel, size_t lds) {          // (as allow_lds of fjsp_kernels.hip: on every launch, per device)
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
// small-batch kernels: waves per workgroup (FJSP_GROUP_WPB=1|2|4 for A/B runs)
static unsigned group_waves_per_block() {
    static const unsigned w = [] { const char *v = getenv("FJSP_GROUP_WPB"); const int x = v ? atoi(v) : 4; return (unsigned)(x == 1 || x == 2 ? x : 4); }();
    return w;
}
template <int V>
static int launch_step_group_v(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                               uint8_t *done, int16_t *trace_km, hipStream_t st) {
    const bool early = group_early(b);
    const unsigned wpb = early ? group_waves_per_block() : 1u, waves = (unsigned)((b.N + 3) / 4);
    const dim3 grid((waves + wpb - 1) / wpb);
#define FJSP_GSTEP(MPC, E) group_allow_lds(&grp::gstep_kernel<V, MPC, E>, wpb * grp::group_lds_bytes<MPC, E>() + group_lds_pad()); hipLaunchKernelGGL((grp::gstep_kernel<V, MPC, E>), grid, dim3(64 * wpb), (wpb * grp::group_lds_bytes<MPC, E>() + group_lds_pad()), st, b, actions, mo, autoreset, state, reward, done, trace_km)
    if (b.MP <= 5) { if (early) { FJSP_GSTEP(5, true); } else { FJSP_GSTEP(5, false); } }
    else { if (early) { FJSP_GSTEP(8, true); } else { FJSP_GSTEP(8, false); } }
#undef FJSP_GSTEP
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_step_group(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                      uint8_t *done, int16_t *trace_km, hipStream_t st) {
    if (!b.grp) return -1;
    if (b.variant == FJSP_VARIANT_SO_FJSSP) return launch_step_group_v<FJSP_VARIANT_SO_FJSSP>(b, actions, mo, autoreset, state, reward, done, trace_km, st);
    if (b.variant == FJSP_VARIANT_MO_FJSSP_DISCRETES)
        return launch_step_group_v<FJSP_VARIANT_MO_FJSSP_DISCRETES>(b, actions, mo, autoreset, state, reward, done, trace_km, st);
    return -1;
}

template <int V>
static int launch_rollout_group_v(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                                  double *state_last, hipStream_t st) {
    const bool early = group_early_rollout(b);
    const unsigned wpb = early ? group_waves_per_block() : 1u, waves = (unsigned)((b.N + 3) / 4);
    const dim3 grid((waves + wpb - 1) / wpb);
    // resident static tables (lean build): when the waves a CU gets -- 256 CUs, one wave per workgroup -- fit its LDS with them
    const unsigned waves_per_cu = (waves + 255u) / 256u;
    const size_t res_lds = 4 * (size_t)(b.MP <= 5 ? grp::res_bytes<5>() : grp::res_bytes<8>());
    const size_t lean_lds = b.MP <= 5 ? grp::group_lds_bytes<5, false>() : grp::group_lds_bytes<8, false>();
    static const int res_forced = [] { const char *v = getenv("FJSP_GROUP_RESIDENT"); return v ? atoi(v) : -1; }();
    const int resident = (!early && (res_forced >= 0 ? res_forced != 0 : waves_per_cu * (lean_lds + res_lds) <= (size_t)152 * 1024)) ? 1 : 0;
    const size_t extra = resident ? res_lds : 0;
#define FJSP_GROLL(MPC, E) group_allow_lds(&grp::grollout_kernel<V, MPC, E>, wpb * grp::group_lds_bytes<MPC, E>() + extra); hipLaunchKernelGGL((grp::grollout_kernel<V, MPC, E>), grid, dim3(64 * wpb), (wpb * grp::group_lds_bytes<MPC, E>() + extra), st, b, actions, mo, T, trace_km, reward, state_last, resident)
    if (b.MP <= 5) { if (early) { FJSP_GROLL(5, true); } else { FJSP_GROLL(5, false); } }
    else { if (early) { FJSP_GROLL(8, true); } else { FJSP_GROLL(8, false); } }
#undef FJSP_GROLL
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_rollout_group(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                         double *state_last, hipStream_t st) {
    if (!b.grp) return -1;
    if (b.variant == FJSP_VARIANT_SO_FJSSP) return launch_rollout_group_v<FJSP_VARIANT_SO_FJSSP>(b, actions, mo, T, trace_km, reward, state_last, st);
    if (b.variant == FJSP_VARIANT_MO_FJSSP_DISCRETES)
        return launch_rollout_group_v<FJSP_VARIANT_MO_FJSSP_DISCRETES>(b, actions, mo, T, trace_km, reward, state_last, st);
    return -1;
}

}  // namespace fjsp
