// HBM layout of an environment batch (shared by the host packer and the kernels).
//
// One WAVEFRONT owns one environment.  Lane l of chunk c owns operation type
// k = 64*c + l (r-major, kind_task_tuple order); lanes 0..M-1 double as the
// machine lanes.  All per-k arrays are therefore stored as rows of KP = 64*KC
// entries so a wave reads each row with one coalesced load per chunk, and the
// (machine x op) matrices are stored machine-major: row m = KP contiguous
// entries (DESIGN.md section "HBM layout").
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
    int64_t reserved;
    double obs_prev[10];     // observation_state v(t)                  (:21)
};
static_assert(sizeof(EnvScalars) == 18 * 8, "EnvScalars must be 18 words");

struct InstHeader { int32_t K, M, R, njobs; };

struct DevBatch {
    int32_t N, n_inst, KC, KP, MP, JP, variant, n_obs, n_static, state_size;
    uint64_t rng_seed;
    // ---- static, per instance -------------------------------------------------
    const InstHeader *ihdr;   // [n_inst]
    const uint32_t *kinfoA;   // [n_inst][KP]  first job of the kind (16) | jobs of the kind (16)
    const uint32_t *kinfoB;   // [n_inst][KP]  stage j (8) | J_r (8) | kind r (8) | flags (8): 1 = last stage, 2 = valid
    const uint32_t *elig;     // [n_inst][KP]  machine_rj_dict[(r,j)] as a bitmask
    uint32_t *fmask;          // [n_inst][KP]  fluid_machine_list as a bitmask (x != 0)
    const uint32_t *efirst4;  // [n_inst][KP]  first four machines of machine_rj_dict[(r,j)] in FILE order (8 bit each)
    const uint16_t *p;        // [n_inst][MP][KP] time_mrj_dict, 0 = ineligible
    const double *x;          // [n_inst][MP][KP] fluid solution (INPUT)
    double *rate;             // [n_inst][MP][KP] fluid_process_rate_rj_dict
    double *arr;              // [n_inst][MP][KP] fluid_unprocessed_rj_arrival_dict
    double *rate_sum;         // [n_inst][KP]     fluid_rate_sum
    double *time_sum;         // [n_inst][KP]     fluid_time_sum
    const int32_t *due;       // [n_inst][JP]     job.due_date
    const uint32_t *jinfo;    // [n_inst][JP]     k of the job's stage 0 (16) | J_r (8)
    const double *sstate;     // [n_inst][8]      static state (MO variant)
    // ---- dynamic, per environment ---------------------------------------------
    EnvScalars *scal;         // [N]
    int32_t *tend;            // [N][MP]  machine.time_end
    int32_t *mjob;            // [N][MP]  machine.job_object (job index)
    uint32_t *jst;            // [N][JP]  job state words
    double *un;               // [N][MP][KP] machine.unprocessed_rj_dict
};

// kernel launchers (fjsp_kernels.hip); all asynchronous on `st`, 0 = launched
int launch_fluid_tables(const DevBatch &b, hipStream_t st);
int launch_reset(const DevBatch &b, const uint8_t *mask, double *state, hipStream_t st);
int launch_step(const DevBatch &b, const uint8_t *actions, int autoreset, double *state, double *reward, uint8_t *done,
                int16_t *trace_km, hipStream_t st);
size_t rollout_lds_bytes(const DevBatch &b);
int launch_rollout(const DevBatch &b, const uint8_t *actions, int T, int16_t *trace_km, double *reward,
                   double *state_last, hipStream_t st);
int launch_read(const DevBatch &b, int64_t *delay, int32_t *makespan, int32_t *completion, int32_t *step_time,
                int32_t *step_count, uint8_t *done, uint32_t *status, hipStream_t st);

}  // namespace fjsp
