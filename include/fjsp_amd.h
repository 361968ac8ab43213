/*
 * include/fjsp_amd.h -- C ABI of libfjsp_amd.so (MI355X / gfx950).
 *
 * The reference (Linshan-Ding/Deep_Reinforcement_Learning_for_FJSP) has no FFI:
 * its boundary is the duck-typed Python protocol  Env(...).reset() / .step(a)
 * plus a handful of attributes.  This header is what a
 * binding for that protocol binds to; each entry point cites the reference
 * function it replaces (paths relative to the reference root).  Plain pointers
 * and sizes only -- no torch / HIP types in the signatures (streams are passed
 * as void*, i.e. a hipStream_t; NULL = the default stream).
 *
 * Conventions
 *   - every function returns 0 on success or a negative FJSP_E_* code;
 *     fjsp_last_error() returns a thread-local message for the last failure;
 *   - "h_" pointers are host memory, "d_" pointers are device (HBM) memory
 *     owned by the caller unless stated otherwise;
 *   - k = koff[r] + j is the r-major operation-type index (kind_task_tuple
 *     order, environments/SO_DFJSP_instance_read.py:25);
 *   - the device entry points are asynchronous on the given stream and do not
 *     synchronise, with one exception: for batches with order arrivals
 *     (instances with more than one order, FJSP_VARIANT_MO_DFJSP)
 *     fjsp_env_step / fjsp_env_rollout synchronise once per step to solve
 *     the fluid LPs of the environments an order reached (SO_FJSSP.py:218-231);
 *     per-env errors are reported in the status array (fjsp_env_read), not
 *     by aborting the batch.
 */
#ifndef FJSP_AMD_H
#define FJSP_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FJSP_ABI_VERSION 1

enum {
    FJSP_OK = 0,
    FJSP_E_ARG = -1,        /* bad argument                                            */
    FJSP_E_IO = -2,         /* file missing / unparsable                               */
    FJSP_E_FORMAT = -3,     /* instance violates a format assumption                   */
    FJSP_E_LP = -4,         /* fluid LP failed                                         */
    FJSP_E_UNSUPPORTED = -5,/* instance outside what the kernels handle (K, M, jobs)   */
    FJSP_E_HIP = -6,        /* HIP runtime error (no device, launch failure, ...)      */
    FJSP_E_STATE = -7       /* call order violated (e.g. fluid solution missing)       */
};

/* per-env status bits written by the kernels (fjsp_env_status) */
enum {
    FJSP_ST_BAD_TASK_RULE = 1,    /* MyError, SO_FJSSP.py:297                          */
    FJSP_ST_BAD_MACHINE_RULE = 2, /* MyError, SO_FJSSP.py:321                          */
    FJSP_ST_STEP_AFTER_DONE = 4,  /* reference: undefined (ValueError on max([]))      */
    FJSP_ST_NO_EVENT = 8          /* reference: ValueError min([]) at SO_FJSSP.py:207  */
};

/* environment variants sharing the SO_FJSSP skeleton */
enum {
    FJSP_VARIANT_SO_FJSSP = 0,          /* environments/SO_FJSSP.py (pair action [6,5], 20-dim state) */
    FJSP_VARIANT_SO_SFJSP = 1,          /* environments/SO_SFJSP.py (flat 20 = 4x5, 18-dim state, makespan)  */
    FJSP_VARIANT_MO_FJSSP_DISCRETES = 2,/* environments/MO_FJSSP_discretes.py (flat 18, 25-dim state)  */
    FJSP_VARIANT_SO_DFJSP = 5,          /* environments/SO_DFJSP.py (agents/DA3C's environment): SO_FJSSP.py over
                                           class_FJSP.py -- job due date = the order's delivery time (class_FJSP.py:229),
                                           Machine.gap_ave without the 1e-18 (:159); same actions, state and kernels */
    FJSP_VARIANT_MO_DFJSP = 4           /* environments/MO_DFJSP_breakdown.py (and MO_DFJSP.py = no breakdown windows):
                                           pair action [12,10], 30-dim state, order arrivals, machine breakdowns,
                                           energy; needs instances with machine data (fjsp_instances_set_dynamic
                                           or a machine_data.csv) */
};

const char *fjsp_last_error(void);
int fjsp_abi_version(void);

/* ------------------------------------------------------------------------- *
 * Host side: instance sets (replaces Data / Instance / FJSP.__init__ / fluid_model)
 * ------------------------------------------------------------------------- */
typedef struct fjsp_instances fjsp_instances;

/* Parameters of the synthetic generator (environments/Instance_generate.py:19-94).
 * The reference generator is unseeded; this one is a counter-based splitmix64
 * stream so instance i of a batch is a pure function of (seed, params). */
typedef struct {
    int32_t R_min, R_max;     /* kinds                     (:42  U{3..12})            */
    int32_t J_min, J_max;     /* ops per kind              (:46  U{3..5})             */
    int32_t M;                /* machines                  (constructor argument)     */
    int32_t p_min, p_max;     /* processing time           (:50  U{40..400})          */
    int32_t N_min, N_max;     /* jobs per kind per order   (:54  U{5..50})            */
    int32_t S;                /* orders                    (constructor argument)     */
    double  DDT;              /* due-date tightness        (constructor argument)     */
    double  t_si_min, t_si_max; /* order inter-arrival     (:58  U(100,200))          */
} fjsp_gen_params;

int  fjsp_instances_create(int32_t n, fjsp_instances **out);
void fjsp_instances_destroy(fjsp_instances *s);
int  fjsp_instances_count(const fjsp_instances *s);

/* Data(path, file_name): environments/SO_DFJSP_instance_read.py:6-89
 * (based_data.csv / process_data.csv / order_data.csv; numbers via the
 * reference's `\d+` extraction, so DDT "0.5" parses to 0). */
int fjsp_instances_load_csv(fjsp_instances *s, int32_t i, const char *path, const char *file_name);
/* Instance(DDT, M, S): environments/Instance_generate.py:24-94, seeded. */
int fjsp_instances_generate(fjsp_instances *s, int32_t i, uint64_t seed, const fjsp_gen_params *prm);
/* Raw arrays (same meaning as the fjsp_instances_get outputs). */
int fjsp_instances_set_raw(fjsp_instances *s, int32_t i, int32_t R, int32_t M, int32_t S,
                           const int32_t *Jr, const int32_t *p /*[K*M] k-major, 0 = ineligible*/,
                           const int32_t *elig_n /*[K]*/, const int32_t *elig_list /*[K*M] file order*/,
                           const int32_t *count /*[S*R]*/, const int32_t *arrive /*[S]*/,
                           const int32_t *delivery /*[S]*/, double ddt);
/* dims[0..5] = R, M, K, S, jobs of order 0, total jobs over all orders */
int fjsp_instances_dims(const fjsp_instances *s, int32_t i, int32_t dims[6]);
/* Any output pointer may be NULL. */
int fjsp_instances_get(const fjsp_instances *s, int32_t i, int32_t *Jr, int32_t *p, int32_t *elig_n,
                       int32_t *elig_list, int32_t *count, int32_t *arrive, int32_t *delivery,
                       double *ddt, double *x /*[K*M] k-major, fluid solution or zeros*/);

/* Dynamic multi-objective folders (environments/MO_DFJSP_instance_read.py:6-108: process_data.csv carries a
 * power column, machine_data.csv the idle power and the breakdown windows of every machine); load_csv reads
 * them when machine_data.csv exists.  dims[0] = 1 if present, dims[1] = total number of breakdown windows.
 * power[K*M] k-major (0 = ineligible), idle_power[M], bk_n[M] windows per machine, bk[2*total] flattened
 * (start, end) pairs machine-major in file order. */
int fjsp_instances_dynamic_dims(const fjsp_instances *s, int32_t i, int32_t dims[2]);
int fjsp_instances_get_dynamic(const fjsp_instances *s, int32_t i, int32_t *power, int32_t *idle_power,
                               int32_t *bk_n, int32_t *bk);
int fjsp_instances_set_dynamic(fjsp_instances *s, int32_t i, const int32_t *power, const int32_t *idle_power,
                               const int32_t *bk_n, const int32_t *bk);

/* FJSP.fluid_model() for the reset-time state (environments/class_FJSSP.py:246-280):
 * solves the fluid LP of instances [first, first+n) with the library's own
 * deterministic vertex simplex on n_threads host threads and stores x. */
int fjsp_instances_solve_fluid(fjsp_instances *s, int32_t first, int32_t n, int32_t n_threads);
/* Override the stored fluid solution (x is an INPUT of the accelerated path). */
int fjsp_instances_set_x(fjsp_instances *s, int32_t i, const double *x /*[K*M] k-major*/);
/* One fluid LP for an arbitrary live state (class_FJSSP.py:246-280):
 * Q[k] = fluid_unprocessed_number_start, n_now[k] = fluid_number. */
int fjsp_fluid_lp(int32_t R, int32_t M, const int32_t *Jr, const int32_t *p /*[K*M]*/,
                  const int32_t *Q /*[K]*/, const int32_t *n_now /*[K]*/,
                  double *x /*[K*M]*/, double *objective);

/* ------------------------------------------------------------------------- *
 * Device side: a batch of N environments resident in HBM
 * ------------------------------------------------------------------------- */
typedef struct fjsp_env fjsp_env;

/* FJSP.__init__ + SO_FJSSP_Environment.__init__ (class_FJSSP.py:151-171,
 * SO_FJSSP.py:14-48) for N envs: env e uses instance (first + e % n_inst).
 * Packs the padded struct-of-arrays, uploads it to `device`, runs the
 * fluid-table kernel (update_fluid_parameter, class_FJSSP.py:282-306) and one
 * internal reset that caches every instance's reset observation; the envs are
 * then left in the "done" state, so a step before fjsp_env_reset is flagged
 * FJSP_ST_STEP_AFTER_DONE (or starts a fresh episode with autoreset).
 * variant: FJSP_VARIANT_*; FJSP_VARIANT_MO_DFJSP needs instances with machine data. */
int  fjsp_env_create(const fjsp_instances *s, int32_t first, int32_t n_inst, int32_t n_envs,
                     int32_t variant, int32_t device, uint64_t rng_seed, fjsp_env **out);
void fjsp_env_destroy(fjsp_env *e);
int  fjsp_env_num_envs(const fjsp_env *e);
int  fjsp_env_state_size(const fjsp_env *e);   /* 20 (SO_FJSSP) / 18 (SO_SFJSP) / 25 (MO_FJSSP_discretes) / 30 (MO_DFJSP) */
int  fjsp_env_device(const fjsp_env *e);

/* reset(): SO_FJSSP.py:51-76.  d_mask (u8[N], nullable): reset only envs with
 * mask != 0.  d_state (f64[N][state_size], nullable) receives the state. */
int fjsp_env_reset(fjsp_env *e, const uint8_t *d_mask, double *d_state, void *stream);

/* step(action): SO_FJSSP.py:168-265.  d_actions u8[N][2] = (task rule, machine
 * rule) indices as the reference's action pair (:171-172); for the flat-action
 * variants (SO_SFJSP, MO_FJSSP_discretes) d_actions[..][0] is the flat action;
 * for the MO variant d_mo (f64[N][4] = w0, w1, completion,
 * tardiness; <=0 = None; nullable) carries step()'s extra arguments
 * (MO_FJSSP_discretes.py:88); for FJSP_VARIANT_MO_DFJSP d_mo is f64[N][4] =
 * reward_policy (0 makespan, 1 tardiness, 2 energy, 3 normalised sum),
 * completion, tardiness, energy_consumption (MO_DFJSP_breakdown.py:189,430-447;
 * NULL = policy 1; any other policy value sets FJSP_ST_BAD_TASK_RULE like the
 * reference's MyError).  Outputs (all nullable): d_state f64[N][S],
 * d_reward f64[N], d_done u8[N].  With d_state == NULL the step skips the
 * observation (state_extract, SO_FJSSP.py:78-97: a third of a step's work) -- for
 * callers that pick rules without looking at the state; the environment notes
 * that its remembered v(t-1) is stale and the next call that does return a state
 * rebuilds it first, so states stay identical to the reference's whatever the
 * mix of calls.  Envs already done are left untouched and get
 * FJSP_ST_STEP_AFTER_DONE unless autoreset != 0, in which case a done env is
 * reset first and the step applies to the fresh episode.  * d_actions must be 2-byte aligned (the kernels read an env's pair as one 16-bit word): FJSP_E_ARG otherwise. */
int fjsp_env_step(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset,
                  double *d_state, double *d_reward, uint8_t *d_done, void *stream);
/* The same step, also reporting what the rule pair resolved to: d_trace_km
 * i16[N][2] (nullable) = the operation type index k (kind_task_tuple order) and
 * the machine m that task_select / machine_select chose (SO_FJSSP.py:173-174),
 * -1 / -1 for an env that did not step. */
int fjsp_env_step_traced(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset,
                         double *d_state, double *d_reward, uint8_t *d_done, int16_t *d_trace_km, void *stream);

/* Asynchronous form of fjsp_env_step for batches with order arrivals (SO_FJSSP.py:218-231, class_MODFJSP.py:240-276:
 * an arrival re-solves the fluid LP on the host).  fjsp_env_step waits for those LPs inside every call; here an env
 * that reaches an arrival PARKS while the rest of the batch keeps stepping: its LP inputs travel to the host, a worker
 * pool solves them, and a later call uploads the solution and finishes the parked step (arrival_kernel).
 * d_ready u8[N]: 1 = this call completed a step of env i (d_state / d_reward / d_done row i are that step's), 0 = the
 * env is parked and its row is untouched.  WHICH ACTION A PARKED STEP APPLIES: the one presented in the call where the
 * env parked (the first call that returns ready = 0 for it) -- that call already ran the step up to the arrival.  Calls
 * made while it stays parked, and the call in which it resumes (ready = 1 again), do not look at its action at all: the
 * resumed row is the result of the action of the parking call.  A caller that records transitions must therefore keep
 * (state, action) of the parking call and pair them with the row that comes back with ready = 1; resampling a
 * stochastic policy for an env that shows ready = 0 has no effect on the environment.  An env's own trajectory is the
 * one fjsp_env_step produces for the sequence of its APPLIED actions (parity is per env); only the interleaving across
 * envs differs.  fjsp_env_arrivals_flush waits for every parked env and finishes its step (rows + ready = 1);
 * it must run before fjsp_env_step / _reset / _rollout / destroy are used on the batch again (they return
 * FJSP_E_STATE while envs are parked; so does fjsp_env_read).  fjsp_env_parked: parked envs as last seen by the host.
 * DEVIATION from the reference's protocol (a single-threaded reset()/step() object): this service
 * runs a dispatcher thread and a pool of LP worker threads inside the library for as long as the batch lives.  They touch
 * only the batch's own staging buffers, never call back into the caller and are joined by fjsp_env_destroy; a caller that
 * forks must do so before the first fjsp_env_step_async.  The blocking fjsp_env_step uses host threads only inside the
 * call (and none at all when the LPs run on the device, fjsp_env_lp_on_device). */
int fjsp_env_step_async(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t autoreset, double *d_state,
                        double *d_reward, uint8_t *d_done, uint8_t *d_ready, void *stream);
int fjsp_env_arrivals_flush(fjsp_env *e, const double *d_mo, double *d_state, double *d_reward, uint8_t *d_done,
                            uint8_t *d_ready, void *stream);
int64_t fjsp_env_parked(const fjsp_env *e);
/* Order-arrival LPs answered from the per-batch memo of (instance, Q, n_now) -> x (a pure function: same bits as a solve). */
int64_t fjsp_env_lp_cache_hits(fjsp_env *e);

/* T fused steps in ONE launch (rule-sweep harnesses, MO_DFJSP.py:481-518 style):
 * d_actions u8[T][N][2]; d_mo as in fjsp_env_step (constant over the T steps);
 * envs that finish early idle (no autoreset).
 * Trace outputs (nullable): d_trace_km i16[T][N][2] = chosen (k, m) or -1,
 * d_reward f64[T][N], d_state_last f64[N][S].  With d_state_last == NULL the
 * fused kernel skips the observation altogether (rule sweeps read makespan /
 * tardiness / energy only); as in fjsp_env_step the next call that returns a
 * state rebuilds the remembered v(t-1) first. */
int fjsp_env_rollout(fjsp_env *e, const uint8_t *d_actions, const double *d_mo, int32_t T, int16_t *d_trace_km,
                     double *d_reward, double *d_state_last, void *stream);

/* Read-back of the attributes agents / harnesses read, device
 * pointers, any may be NULL: delay_time_sum i64[N], makespan = max machine
 * time_end i32[N], completion_time i32[N], step_time i32[N], step_count i32[N],
 * done u8[N], status u32[N] (FJSP_ST_* bits, sticky until reset). */
int fjsp_env_read(fjsp_env *e, int64_t *d_delay_time_sum, int32_t *d_makespan, int32_t *d_completion,
                  int32_t *d_step_time, int32_t *d_step_count, uint8_t *d_done, uint32_t *d_status,
                  void *stream);
/* machine_dict[m].time_end: i32[N][M_max] (SO_FJSSP.py:426). */
int fjsp_env_machine_time_end(fjsp_env *e, int32_t *d_tend, int32_t m_stride, void *stream);
/* energy_consumption i64[N] of a FJSP_VARIANT_MO_DFJSP batch (MO_DFJSP_breakdown.py:253-256). */
int fjsp_env_energy(fjsp_env *e, int64_t *d_energy, void *stream);
/* fluid tables of env i copied to host (tests): rate/arr [K*M] k-major, rate_sum/time_sum [K]. */
int fjsp_env_fluid_tables(fjsp_env *e, int32_t i, double *h_rate, double *h_arr,
                          double *h_rate_sum, double *h_time_sum);
/* Test hook (host build of csrc/fjsp_pyset.h, the code machine_select runs on the
 * device): iteration order of list(set(idle) & set(machines)) as CPython 3.10
 * produces it (SO_FJSSP.py:302-303).  idle_mask bit m = machine m idle;
 * machines[n] in tuple order (ascending != 0: inserted in ascending order).
 * Writes the order to out[], returns its length (or a negative error). */
int fjsp_pyset_and_order(uint32_t idle_mask, const int32_t *machines, int32_t n, int32_t ascending, int32_t *out);
/* HBM bytes the step kernel reads+writes per env-step (algorithmic). */
int64_t fjsp_env_step_bytes(const fjsp_env *e);
/* Which kernel family steps this batch: 0 = one wavefront per environment (csrc/fjsp_kernels.hip: every variant and
 * shape), 1 = one 16-lane row per environment (csrc/fjsp_group.hip: SO_FJSSP / SO_DFJSP / MO_FJSSP_discretes batches of
 * one job per kind, one order, <= 64 operation types, <= 8 machines, <= 15 jobs -- the reference's 10x5 and Brandimarte
 * shapes).  Same results either way; the environment variable FJSP_STEP_IMPL=wave at create time forces 0. */
int fjsp_env_kernel_family(const fjsp_env *e);
/* Order arrivals (SO_FJSSP.py:218-231) re-solve the fluid LP on the host, one LP per arriving env, spread
 * over n_threads host threads (0 = default: min(host cores, 16)).  fjsp_env_lp_solves: LPs solved so far. */
int fjsp_env_set_lp_threads(fjsp_env *e, int32_t n_threads);
int64_t fjsp_env_lp_solves(const fjsp_env *e);
/* Where the order-arrival LPs of this batch are solved: 1 = on the device (csrc/fjsp_lp_device.hip: the host simplex of
 * csrc/fjsp_lp.cpp restated pivot for pivot, one workgroup per parked environment, tableau in LDS; fjsp_env_step then never
 * synchronises), 0 = on the host.  Same x either way, bit for bit.  Chosen at create time: the device when the largest tableau
 * of the batch fits a CU's LDS (and is at most 512 columns wide) and the batch has 16384 environments or more -- a single LP is
 * ~6x slower on the device than on a host core, 256 run at once: below that size the host service with its cache of solved
 * LPs is as fast or faster (DESIGN.md has the measurements).  FJSP_LP_IMPL=device / host at create time overrides the size rule.
 * fjsp_env_lp_device_solve (test hook): the device solver on one LP of env's instance -- Q[K], n_now[K] as
 * class_FJSSP.py:234-237 builds them -- x f64[K*M] (k-major) to the host; the batch must have no parked environments. */
int fjsp_env_lp_on_device(const fjsp_env *e);
/* Pivots the device simplex has executed so far, all LPs together (0 when the batch keeps the host service; synchronises). */
int64_t fjsp_env_lp_device_pivots(const fjsp_env *e);
int fjsp_env_lp_device_solve(fjsp_env *e, int32_t env, const int32_t *Q, const int32_t *n_now, double *x);

/* ------------------------------------------------------------------------- *
 * Rollout buffer (on-policy Replay_Buffer, agents/MPPPO/Buffer.py:7-58) in HBM
 * ------------------------------------------------------------------------- */
typedef struct fjsp_rollout fjsp_rollout;

/* Capacity T steps x N envs x state_size; storage f32 like Buffer.py:41-45. */
int  fjsp_rollout_create(int32_t T, int32_t N, int32_t state_size, int32_t device, fjsp_rollout **out);
void fjsp_rollout_destroy(fjsp_rollout *b);
/* add_experience (Buffer.py:19-28): converts the f64 env outputs of one
 * batched step to f32 rows at the write cursor t. `d_active` u8[N] (nullable)
 * = envs that were not done before the step (valid-row mask). */
int fjsp_rollout_append(fjsp_rollout *b, const double *d_state, const uint8_t *d_actions,
                        const double *d_reward, const double *d_next_state, const uint8_t *d_done,
                        const uint8_t *d_active, void *stream);
/* calculate_discounted_returns (agents/MPPPO/MPPPO.py:301-312): reverse scan
 * G_t = r_t + gamma * G_{t+1} per env over valid rows, f64 scan -> f32. */
int fjsp_rollout_returns(fjsp_rollout *b, double gamma, void *stream);
/* The same scan followed by the per-episode normalisation of MPPPO.py:258-261 (normalized: (G - min) / (max - min +
 * 1e-8) over the env's valid rows; standardized: (G - mean) / (unbiased std + 1e-8)) in one launch; d_out
 * f32[len][N] receives the result (0 in rows that are not valid), the buffer's returns the raw scan. */
int fjsp_rollout_returns_normalised(fjsp_rollout *b, double gamma, int32_t normalized, int32_t standardized, float *d_out, void *stream);
/* clear(): Buffer.py:53-55 */
int fjsp_rollout_clear(fjsp_rollout *b);
int fjsp_rollout_len(const fjsp_rollout *b);
/* Device pointers into the buffer (wrapped zero-copy by the PyTorch side):
 * which = 0 states f32[T][N][S], 1 actions f32[T][N][2], 2 rewards f32[T][N],
 * 3 next_states f32[T][N][S], 4 dones f32[T][N], 5 valid f32[T][N], 6 returns f32[T][N]. */
void *fjsp_rollout_ptr(fjsp_rollout *b, int32_t which);

/* The actor of the on-policy agents (ActorNet, agents/MPPPO/MPPPO.py:31-48) at the size of BASELINE config 3:
 * state_size -> 128 -> 128 -> n_actions, Linear + ReLU twice, Linear, softmax.  Device pointers to the f32
 * parameters in torch's layout (weight[out][in] row-major): what nn.Linear holds.  state_size <= 32, hidden == 128,
 * n_actions <= 32. */
typedef struct fjsp_actor_params {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    int32_t state_size, hidden, n_actions;
} fjsp_actor_params;

/* ActorNet.forward (MPPPO.py:44-48) for n states: d_state f64[n][state_size] (converted with `.float()` like
 * MPPPO.py:274) -> d_probs f32[n][n_actions].  The same device code the fused rollout below evaluates in place. */
int fjsp_actor_forward(const fjsp_actor_params *actor, const double *d_state, int32_t n, float *d_probs, void *stream);

/* run_one_policy_network's rollout loop (MPPPO.py:245-252: pick_action_and_log_prob, env.step, save_experience)
 * for T vector steps in ONE launch: the actor is evaluated inside the environment kernel, actions are drawn from
 * the stream of fjsp_policy_sample (same *d_seed, counter = step index: the same actions as the per-step path
 * actor -> fjsp_policy_sample -> fjsp_env_step, bit for bit), and every step's row goes straight into `buf`
 * (as fjsp_rollout_append writes it; rows of envs that had finished are marked invalid, buf's length becomes T).
 * d_state_in f64[N][S]: the states the rollout starts from (what fjsp_env_reset returned); d_flat_actions /
 * d_log_prob f32[T][N]: the sampled action index and its log-probability; d_state_last f64[N][S] receives every
 * env's latest state.  d_mo as in fjsp_env_step.  Single-order batches of at most 64 operation types;
 * FJSP_E_UNSUPPORTED otherwise (callers fall back to the per-step loop). */
int fjsp_env_rollout_policy(fjsp_env *e, fjsp_rollout *buf, const fjsp_actor_params *actor, const float *d_epsilon,
                            const uint64_t *d_seed, int32_t pair_div, int32_t T, const double *d_mo, const double *d_state_in,
                            float *d_flat_actions, float *d_log_prob, double *d_state_last, void *stream);

/* pick_action_and_log_prob (agents/MPPPO/MPPPO.py:272-284) for one vector step in ONE launch: samples
 * Categorical(d_probs[env]) (f32[n][n_actions], the actor's softmax output), applies the epsilon-random
 * override (*d_epsilon, device scalar so that captured graphs can change it), and writes the flat action
 * (d_action f32[n]), its log-probability (d_log_prob f32[n]) and the action in the environment's encoding
 * (d_pair u8[n][2] = (a / pair_div, a % pair_div), or (a, 0) when pair_div == 0).  Random numbers are a
 * counter-based function of (*d_seed, counter, env). */
int fjsp_policy_sample(const float *d_probs, int32_t n, int32_t n_actions, int32_t pair_div, const float *d_epsilon,
                       const uint64_t *d_seed, uint64_t counter, uint8_t *d_pair, float *d_action, float *d_log_prob,
                       void *stream);

/* ---- fused pieces of the clipped-PPO learning iteration (agents/MPPPO/MPPPO.py:314-370; csrc/fjsp_ppo.hip).
 * All pointers are device pointers, f32; *d_count is the (global) number of samples the losses average over.
 *
 * fjsp_ppo_actor_loss: logits f32[n][n_actions] of the new policy, d_actions f32[n] (flat action index),
 * d_old_log_prob / d_advantages f32[n] -> *d_loss = -sum(min(A r, A clip(r, 1 - eps, 1 + eps))) / count with
 * r = exp(log_softmax(logits)[action]) / (exp(old) + 1e-8) (:325-352), and d_dlogits = d loss / d logits with
 * autograd's conventions (minimum halves ties, clamp passes the gradient on the closed interval).
 * d_partial: f32 scratch of fjsp_ppo_partials(n) entries.
 * fjsp_ppo_critic_loss: *d_loss = sum((value - returns)^2) / count (F.mse_loss, :317-318), d_dvalue = its gradient.
 * fjsp_relu_bwd_bias: d_dh f32[n][width] <- d_dh * (d_h > 0) in place (d_h == NULL: no mask) and d_bias_grad
 * f32[width] = its column sums, through d_partial f32[n_partial_rows][width] (two-stage, deterministic).
 * fjsp_adam_clip_step: clip_grad_norm_(max_norm; <= 0: none) followed by one Adam step (torch.optim.Adam, no weight
 * decay) over flat buffers of n parameters; *d_step is the step count kept on the device (f32), d_scratch64 f32[64]. */
int fjsp_ppo_partials(int32_t n);
int fjsp_ppo_actor_loss(const float *d_logits, const float *d_actions, const float *d_old_log_prob, const float *d_advantages, int32_t n,
                        int32_t n_actions, float clip_epsilon, const float *d_count, float *d_dlogits, float *d_partial, float *d_loss,
                        void *stream);
int fjsp_ppo_critic_loss(const float *d_value, const float *d_returns, int32_t n, const float *d_count, float *d_dvalue, float *d_partial,
                         float *d_loss, void *stream);
int fjsp_relu_bwd_bias(float *d_dh, const float *d_h, int32_t n, int32_t width, float *d_partial, int32_t n_partial_rows, float *d_bias_grad,
                       void *stream);
int fjsp_adam_clip_step(float *d_params, const float *d_grads, float *d_exp_avg, float *d_exp_avg_sq, int32_t n, float max_norm, float lr,
                        float beta1, float beta2, float eps, float *d_step, float *d_scratch64, void *stream);

/* ---- one launch per network and learning iteration: forward + loss + backward on the f32 matrix cores
 * (csrc/fjsp_mlp_train.hip), for the reference's Linear-ReLU-Linear-ReLU-Linear networks with 128 hidden units
 * (agents/MPPPO/MPPPO.py:27-70 actor / critic; the iteration is :314-370).  d_params: the network's parameters in ONE
 * flat f32 buffer, order W1[128][S] b1[128] W2[128][128] b2[128] W3[n_out][128] b3[n_out] (16-byte aligned);
 * d_x f32[n][S].  mode 0 (actor): d_aux0 = actions (flat index as f32), d_aux1 = old log-probabilities, d_aux2 =
 * advantages; loss and gradient as fjsp_ppo_actor_loss.  mode 1 (critic, n_out == 1): d_aux0 = returns; loss as
 * fjsp_ppo_critic_loss.  Outputs: d_grad f32[numel] = d loss / d params (same order as d_params), *d_loss.
 * Scratch: d_partial f32[n_groups][numel], d_loss_partial f32[n_groups], n_groups = fjsp_mlp_train_groups(n).
 * FJSP_E_UNSUPPORTED unless state_size <= 31, hidden == 128, n_out <= 32.
 * fjsp_mlp_train_step: the pass followed by the optimiser step of fjsp_adam_clip_step on d_grad, in three launches
 * (pass, gradient finish + squared norm + step count, clip + Adam); d_sumsq_partial: f32 scratch of (numel + 63) / 64
 * entries.  Single-process form: with several GPUs call fjsp_mlp_train_pass, all-reduce d_grad, fjsp_adam_clip_step. */
int fjsp_mlp_train_groups(int32_t n);
int fjsp_mlp_train_step(int32_t mode, float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden, int32_t n_out,
                        const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count, float clip_epsilon,
                        float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss, float *d_exp_avg,
                        float *d_exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float *d_step,
                        float *d_sumsq_partial, void *stream);
int fjsp_mlp_train_pass(int32_t mode, const float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden,
                        int32_t n_out, const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count,
                        float clip_epsilon, float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss,
                        void *stream);

/* fjsp_mlp_train_step for the critic (mode 1) that also hands out what its forward computed: d_values_out f32[n] = V(s) of
 * every sample under the parameters BEFORE this step's update -- the values the advantages of the round are built from
 * (MPPPO.py:263 returns - critic(states)), so the first critic iteration of a learning round replaces the separate
 * forward pass. */
int fjsp_mlp_train_step_values(int32_t mode, float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden,
                               int32_t n_out, const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count,
                               float clip_epsilon, float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad,
                               float *d_loss, float *d_exp_avg, float *d_exp_avg_sq, float max_norm, float lr, float beta1,
                               float beta2, float eps, float *d_step, float *d_sumsq_partial, float *d_values_out, void *stream);

/* ------------------------------------------------------------------------- *
 * Action sampling of the HMPSAC policy networks in one launch
 * (agents/HMPSAC/SAC_Discrete.py:277-284 pick_lower_action, :248-254 pick_action;
 *  A3C_v5.1.py:35-75 TaskPolicyNet / MachinePolicyNet)
 * ------------------------------------------------------------------------- */
/* a_task ~ Categorical(softmax(task(state.float()))), then -- if machine_layers > 0 --
 * a_machine ~ Categorical(softmax(machine(cat(state.float(), a_task)))) for `rows` states f64[rows][state_size].
 * A network is Linear-ReLU-...-Linear: n linear layers (1..6), dims[n + 1] widths (each <= 256, outputs <= 64,
 * dims[0] = state_size, resp. state_size + 1), weights[l] f32[dims[l]][dims[l+1]] row-major -- the TRANSPOSE of
 * torch.nn.Linear.weight, so that the threads of a wave read consecutive words -- and biases[l] f32[dims[l+1]], device pointers.  Randomness: a counter-based stream per row -- splitmix64(seed, row,
 * d_draws[row]) -- whose draw counters u32[rows] live in device memory and advance with every call, so the launch can be
 * replayed from a HIP graph.  d_p_task / d_p_machine (nullable): the f32 probabilities the actions were drawn from.
 * d_pair (nullable, needs the machine network): u8[rows][2] = (a_task, a_machine), the action pair as fjsp_env_step takes it,
 * written for the rows with d_select[row] == which (d_select NULL: every row) -- pick_lower_action's "env e follows lower
 * policy which[e]" without the gather / where launches around it.
 * Other shapes: FJSP_E_UNSUPPORTED (the caller keeps the library path). */
int fjsp_policy_pair_sample(int32_t task_layers, const int32_t *task_dims, const float *const *task_w, const float *const *task_b,
                            int32_t machine_layers, const int32_t *machine_dims, const float *const *machine_w,
                            const float *const *machine_b, const double *d_state, int32_t rows, int32_t state_size, uint64_t seed,
                            uint32_t *d_draws, int64_t *d_a_task, int64_t *d_a_machine, float *d_p_task, float *d_p_machine,
                            uint8_t *d_pair, const int64_t *d_select, int32_t which, void *stream);

#ifdef __cplusplus
}
#endif
#endif
