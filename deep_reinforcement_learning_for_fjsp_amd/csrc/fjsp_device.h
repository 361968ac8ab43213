// HBM layout of an environment batch (shared by the host packer and the kernels).
//
// One WAVEFRONT owns one environment.  Lane l of chunk c owns operation type
// k = 64*c + l (r-major, kind_task_tuple order); lanes 0..M-1 double as the
// machine lanes.
//
// Two slabs, one hipMalloc each:
//   * the INSTANCE slab: one contiguous record per problem instance (static,
//     shared by every environment that plays it);
//   * the ENV slab: one contiguous record per environment (dynamic state).
// Inside a record every per-k array is a row of KP = 64*KC entries (one
// coalesced load per chunk); the (machine x op) matrices are op-major (the MP
// entries of one operation type are contiguous: the decision gathers a column).  A wave therefore touches two address
// ranges per step -- its instance record and its own env record -- instead of
// a dozen unrelated arrays (fewer TLB entries and DRAM pages per wave), and all
// of its start-up loads can be issued before the first wait.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fjsp {

constexpr int kWave = 64;
constexpr int kMaxM = 32;          // machines fit a 32-bit idle mask
constexpr int kMaxKC = 4;          // K <= 256 operation types
constexpr uint32_t kNoSeq = 0xFFFFFFu;

// jstate word per job: next unassigned stage (8 bit) | FIFO sequence (24 bit, kNoSeq = not waiting)
__host__ __device__ inline uint32_t jst_pack(uint32_t seq, uint32_t next_j) { return (seq << 8) | next_j; }
// Where operation type k's byte sits in Layout::e_asg: inside each block of 64 the bytes of k, k + 16, k + 32, k + 48 share
// one 32-bit word, so a lane of the group kernels (fjsp_group.hip: lane l of a 16-lane row owns k = 16 s + l) fetches the
// bytes of its four slots with one load
__host__ __device__ inline uint32_t asg_pos(uint32_t k) { return (k & ~63u) | ((k & 15u) << 2) | ((k >> 4) & 3u); }

// Dynamic per-environment scalars, 18 x 8-byte words (loaded by lanes 0..17).
struct EnvScalars {
    int32_t t;               // self.step_time                          (SO_FJSSP.py:18)
    int32_t step_count;      // self.step_count                         (:17)
    int32_t done;            // self.done                               (:26)
    int32_t n_unassigned;    // sum(len(kind.job_unprocessed_list))     (:247)
    uint32_t status;         // FJSP_ST_* bits
    uint32_t seq_ctr;        // FIFO append counter (job_now_list order)
    uint32_t rng_calls;      // random.choice call counter              (:295,319)
    uint32_t busy;           // bit m = machine_dict[m].state           (class_FJSSP.py:120)
    int32_t completion;      // completion_time                         (MO_FJSSP_discretes.py:122)
    int32_t completion_last;
    int64_t tard_done;       // delay_time_sum_processed                (SO_FJSSP.py:45)
    int64_t delay_sum;       // delay_time_sum == delay_time_sum_last after a step (:44,263)
    int32_t t_arr;           // self.order_arrive_time                  (:19)
    int16_t next_order;      // index of the first order not yet arrived (order_object_list, :53-56)
    int8_t pending;          // 1 = the event loop stopped at an order arrival and waits for its fluid LP
    int8_t obs_stale;        // 1 = obs_prev is not the observation of the current state (steps that handed no state back)
    double obs_prev[10];     // observation_state v(t)                  (:21)
};
static_assert(sizeof(EnvScalars) == 18 * 8, "EnvScalars must be 18 words");

// MO_DFJSP extras of the env record (Layout::e_dyn)
struct DynScalars {
    int64_t energy;          // self.energy_consumption                 (MO_DFJSP_breakdown.py:40,253-256)
    int64_t energy_last;     // self.energy_consumption_last            (:41)
    double obs_hi[6];        // observation_state[10..14] (EnvScalars::obs_prev holds [0..9])
};

struct InstHeader { int32_t K, M, R, njobs; };

// byte offsets inside the two record types
struct Layout {
    // instance record: InstHeader at 0
    uint32_t i_stride;
    uint32_t i_kA;      // u32[KP]  first job of the kind (16) | jobs of the kind (16)
    uint32_t i_kB;      // u32[KP]  stage j (8) | J_r (8) | kind r (8) | flags (8): 1 = last stage, 2 = valid
    uint32_t i_elig;    // u32[KP]  machine_rj_dict[(r,j)] as a bit mask
    uint32_t i_fmask;   // u32[KP]  fluid_machine_list as a bit mask (x != 0)           [written by fluid_tables_kernel]
    uint32_t i_f4;      // u32[KP]  first four machines of machine_rj_dict[(r,j)] in FILE order, 8 bit each
    uint32_t i_rsum;    // f64[KP]  fluid_rate_sum                                       [fluid_tables_kernel]
    uint32_t i_tsum;    // f64[KP]  fluid_time_sum                                       [fluid_tables_kernel]
    uint32_t i_due;     // i32[JP]  job.due_date
    uint32_t i_jinfo;   // u32[JP]  k of the job's stage 0 (16) | J_r (8)
    // (machine x op) matrices are OP-MAJOR: entry (k, m) at k*MP + m, so the column of one operation type
    // -- what machine_select gathers once k is chosen -- is MP contiguous entries
    uint32_t i_p;       // u16[KP][MP] time_mrj_dict, 0 = ineligible
    uint32_t i_x;       // f64[KP][MP] fluid solution (INPUT)
    uint32_t i_col;     // f64[KP][MP][2] {fluid_unprocessed_rj_arrival_dict, fluid_process_rate_rj_dict} [fluid_tables_kernel]
    uint32_t i_ss;      // f64[8]   static state (MO variant); [7] = fluid_completed_time of the reset-time LP
    uint32_t i_obs0;    // f64[16]  observation of the reset state (written by reset_kernel, read by the autoreset path)
    uint32_t i_oarr;    // i32[SP]  time_arrive_s_dict                                   (multi-order batches)
    uint32_t i_ocnt;    // u16[SP][RP] count_sr_dict                                     (multi-order batches)
    // MO_DFJSP batches only (MO_DFJSP_instance_read.py:56-93)
    uint32_t i_pw;      // u16[KP][MP] power_mrj_dict (op-major like i_p)
    uint32_t i_ipw;     // i32[MP]  power_m_dict (idle power)
    uint32_t i_bkoff;   // u16[MP+1] first breakdown window of machine m in i_bk
    uint32_t i_bk;      // i32[BP][2] breakdown windows (start, end), machine-major, file order
    // group-kernel batches only (DevBatch::grp): the static per-operation rows once more, packed for one 16-lane row per
    // environment -- slot s (k = 16 s + l): 16 x uint4 {kB, elig | fmask << 8, due date of the kind's job, 0}, then
    // 16 x double2 {fluid_rate_sum, fluid_time_sum}: two 16-byte loads per lane and slot  [written by fluid_tables_kernel];
    // behind the four slots one line of per-lane words (machine / job / instance: fjsp_env.hip) and the jobs' due dates
    uint32_t i_op;
    // ... and the {arrival, rate} table once more MACHINE-major, f64[MP][64][2]: Machine.gap_ave reads whole machines, and in
    // large batches only the candidate ones (fjsp_group.hip g_gap_ave_lean: 5 lines per candidate instead of the whole table)
    uint32_t i_colm;
    uint32_t i_op8;          // row-kernel batches: {stage | J_r | kind word, elig | fmask} of every operation type, 8 B each, contiguous
                             // (the large-batch build's operation words: 40 types = 2.5 lines instead of the 5 of the 16-byte slots)
    // env record: EnvScalars at 0
    uint32_t e_stride;
    uint32_t e_tend;    // i32[MP]  machine.time_end
    uint32_t e_mjob;    // i32[MP]  machine.job_object (job index)
    uint32_t e_jst;     // u32[JP]  job state words
    uint32_t e_un;      // f64[KP][MP] machine.unprocessed_rj_dict (op-major); batches with several jobs per kind
    uint32_t e_asg;     // u8[KP]   single-job batches: the machine operation type k was assigned to (0xFF: not yet).  There every
                        //          type is dispatched once, so unprocessed[k][m] = arrival[k][m] - (assigned[k] == m): no matrix
    // multi-order batches only: the fluid tables change at every order arrival, so they live per environment
    uint32_t e_q0;      // u32[KP]  fluid_unprocessed_number_start
    uint32_t e_fmask;   // u32[KP]
    uint32_t e_rsum;    // f64[KP]
    uint32_t e_tsum;    // f64[KP]
    uint32_t e_col;     // f64[KP][MP][2]
    uint32_t e_dyn;     // MO_DFJSP batches only: DynScalars, then i32[MP] time_end of the machine's last task (-1 = none)
    uint32_t e_stats;   // u32[KP][16] update_parameter's per-(r, j) statistics as the last kernel left them (batches with several
                        // jobs per kind only: there they are a walk over the kind's jobs, too long to redo at every step entry)
    uint32_t e_lpq;     // u16[2][KP] LP inputs of the pending arrival: Q[k], n_now[k]; then i16[2] stashed (k, m) of the step
};

// Offsets that follow from the padded sizes alone (the host's layout code in fjsp_env.hip places these fields first,
// in this order, and fjsp_env_create verifies the two agree): the kernels take them from here -- compile-time for the
// per-k rows, two shifts for the head of the env record -- instead of fetching them from the kernel arguments, and the
// constants fold into the loads' immediate offsets.
struct FixedOffsets {
    static constexpr uint32_t kInstHeader = 64, kEnvScalars = 192;
    static constexpr __host__ __device__ uint32_t i_kA(uint32_t KP) { (void)KP; return kInstHeader; }
    static constexpr __host__ __device__ uint32_t i_kB(uint32_t KP) { return kInstHeader + 4 * KP; }
    static constexpr __host__ __device__ uint32_t i_elig(uint32_t KP) { return kInstHeader + 8 * KP; }
    static constexpr __host__ __device__ uint32_t i_fmask(uint32_t KP) { return kInstHeader + 12 * KP; }
    static constexpr __host__ __device__ uint32_t i_f4(uint32_t KP) { return kInstHeader + 16 * KP; }
    static constexpr __host__ __device__ uint32_t i_rsum(uint32_t KP) { return kInstHeader + 20 * KP; }
    static constexpr __host__ __device__ uint32_t i_tsum(uint32_t KP) { return kInstHeader + 28 * KP; }
    static constexpr __host__ __device__ uint32_t i_due(uint32_t KP) { return kInstHeader + 36 * KP; }
    static constexpr __host__ __device__ uint32_t e_tend() { return kEnvScalars; }
    static constexpr __host__ __device__ uint32_t e_mjob(uint32_t MP) { return kEnvScalars + 4 * MP; }
    static constexpr __host__ __device__ uint32_t e_jst(uint32_t MP) { return kEnvScalars + 8 * MP; }
    static constexpr __host__ __device__ uint32_t e_un(uint32_t MP, uint32_t JP) { return (kEnvScalars + 8 * MP + 4 * JP + 7u) & ~7u; }   // JP is a multiple of 16
    static constexpr __host__ __device__ uint32_t e_asg(uint32_t MP, uint32_t JP, uint32_t KP, bool single_job) {
        return e_un(MP, JP) + (single_job ? 8u : 8u * MP * KP);
    }
    // single-order, non-dynamic batches: the env record ends with the assigned-machine bytes (single-job) or the 64-byte
    // aligned statistics rows behind them
    static constexpr __host__ __device__ uint32_t e_stride_plain(uint32_t MP, uint32_t JP, uint32_t KP, bool single_job) {
        const uint32_t end = single_job ? e_asg(MP, JP, KP, true) + KP : ((e_asg(MP, JP, KP, false) + KP + 63u) & ~63u) + 64u * KP;
        return (end + 127u) & ~127u;       // whole 128-byte lines
    }
};

struct DevBatch {
    int32_t N, n_inst, KC, KP, MP, JP, variant, n_obs, n_static, state_size;
    int32_t mord, SP, RP;    // multi-order batch (S > 1): order / kind paddings of i_oarr, i_ocnt
    int32_t single_job;      // every kind of every instance has one job and there is one order (10x5, the Brandimarte sets):
                             // per-(r, j) lists have at most one member, job index == kind index
    int32_t jcap;            // jobs of the largest instance, rounded up to 16: lanes beyond it load no job words
    int32_t grp;             // 1 = the batch fits the group kernels (fjsp_group.hip): one job per kind, one order, <= 64 operation
                             // types, <= 8 machines, <= 15 jobs, SO_FJSSP or MO_FJSSP_discretes
    int32_t kmax;            // operation types of the largest instance
    int kenv_first;          // large batches fetch kenv[env] before their operation words (1; FJSP_GROUP_KENV=0 for A/B runs: three
                             // slots unconditionally, the fourth after K is known -- one dependent fetch less, 1-2 lines more: same speed)
    const uint8_t *kenv;     // row-kernel batches: operation types of the instance every environment plays, u8[N] (the large-batch
                             // kernels read it first and request no operation rows beyond it)
    uint32_t *pending_count; // [0] number of envs parked at an order arrival by the last launch, [1 + slot] their env ids
    uint16_t *lp_in;         // [slot][2][KP] LP inputs (Q, n_now) of the parked env in that slot (written when it parks)
    double *lp_x;            // [slot][KP][MP] fluid solution of that LP (uploaded by the host service, read by arrival_kernel)
    uint64_t rng_seed;
    unsigned char *inst;     // [n_inst] instance records
    unsigned char *envs;     // [N] env records
    Layout L;
};

template <class T>
__host__ __device__ inline T *inst_ptr(const DevBatch &b, int inst, uint32_t off) {
    return reinterpret_cast<T *>(b.inst + (size_t)inst * b.L.i_stride + off);
}
template <class T>
__host__ __device__ inline T *env_ptr(const DevBatch &b, int env, uint32_t off) {
    return reinterpret_cast<T *>(b.envs + (size_t)env * b.L.e_stride + off);
}

// kernel launchers (fjsp_kernels.hip); all asynchronous on `st`, 0 = launched
int launch_fluid_tables(const DevBatch &b, hipStream_t st);
int launch_reset(const DevBatch &b, const uint8_t *mask, double *state, hipStream_t st);
// ready (nullable, multi-order batches): asynchronous arrival service, see fjsp_env_step_async
int launch_step(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                uint8_t *done, int16_t *trace_km, hipStream_t st, uint8_t *ready = nullptr);
// the same step by the group kernels (fjsp_group.hip; DevBatch::grp batches only)
int launch_step_group(const DevBatch &b, const uint8_t *actions, const double *mo, int autoreset, double *state, double *reward,
                      uint8_t *done, int16_t *trace_km, hipStream_t st);
int launch_rollout_group(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                         double *state_last, hipStream_t st);
size_t rollout_lds_bytes(const DevBatch &b);
size_t step_lds_bytes(const DevBatch &b);     // dynamic LDS of one reset / step / arrival workgroup
int launch_rollout(const DevBatch &b, const uint8_t *actions, const double *mo, int T, int16_t *trace_km, double *reward,
                   double *state_last, hipStream_t st);
// multi-order: resume the envs whose pending LP has been solved (x in e_xin)
// ids u32[n_pending] / x_list f64[n_pending][KP][MP]: the parked envs and their LP solutions (device)
int launch_arrival(const DevBatch &b, const double *mo, int n_pending, const uint32_t *ids, const double *x_list, double *state,
                   double *reward, uint8_t *done, int16_t *trace_km, hipStream_t st, uint8_t *ready = nullptr, bool mark_resumed = false,
                   const uint32_t *n_dev = nullptr);     // n_dev: read the count on the device (the grid then covers the batch)
// the order-arrival LPs on the device (fjsp_lp_device.hip): one workgroup per parked environment, x into lp_x[slot]
size_t lp_device_lds_bytes(int K, int M, int nx, int R, int MP);
int lp_device_max_columns();      // widest tableau the device simplex takes (its objective row lives in registers)
int launch_lp_device(const DevBatch &b, const uint32_t *count_dev, int count_host, const uint32_t *ids, const uint16_t *lp_in, double *lp_x,
                     uint32_t *err, unsigned long long *solved, size_t lds, hipStream_t st);
// policy inside the launch (fjsp_policy.h)
struct ActorParams;
struct PolicyRolloutIO;
size_t policy_rollout_lds_bytes(const DevBatch &b, int S);
int launch_actor_forward(const ActorParams &ap, const double *state, int n, float *probs, hipStream_t st);
int launch_rollout_policy(const DevBatch &b, const ActorParams &ap, const PolicyRolloutIO &io, const double *mo, int T, hipStream_t st);
int launch_read(const DevBatch &b, int64_t *delay, int32_t *makespan, int32_t *completion, int32_t *step_time,
                int32_t *step_count, uint8_t *done, uint32_t *status, hipStream_t st);

}  // namespace fjsp
