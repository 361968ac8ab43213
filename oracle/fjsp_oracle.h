/*
 * oracle/fjsp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement (plain C, one env, Python-object semantics kept as
 * explicit ordered lists) of the reference's rule-dispatch discrete-event
 * environment:
 *     environments/SO_FJSSP.py:51-389        (reset / step / rules / state)
 *     environments/class_FJSSP.py:13-306     (object model, due dates, fluid parameters)
 *     environments/SO_SFJSP.py, MO_FJSSP_discretes.py, SO_DFJSP.py (+ class_FJSP.py),
 *     MO_DFJSP_breakdown.py (+ class_MODFJSP.py)   (the variants of fjo_create)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product path (deep_reinforcement_learning_for_fjsp_amd)
 * never does; it fails loudly when the HIP library is missing.
 *
 * Parity status: pinned downstream of the fluid LP solution x (golden vectors
 * generated from the reference itself, tests/golden/make_golden.py: 3 061
 * episodes; plus 19 752 random-shape episodes, fuzz_oracle_vs_reference.py);
 * "parity unpinned" AT the LP boundary (docplex/CPLEX absent, optimum
 * non-unique) -- x is an input here.
 */
#ifndef FJSP_ORACLE_H
#define FJSP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fjo_env fjo_env;

/* Instance in the array form of SURVEY.md Appendix A.  k = koff[r] + j is the
 * r-major operation-type index (kind_task_tuple order, SO_DFJSP_instance_read.py:25). */
typedef struct {
    int R, M, K, S;
    const int *Jr;        /* [R]   ops per kind (len(task_r_dict[r]))                   */
    const int *p;         /* [K*M] p[k*M+m] = time_mrj_dict[m][(r,j)], 0 = ineligible   */
    const int *elig_n;    /* [K]   len(machine_rj_dict[(r,j)])                          */
    const int *elig_list; /* [K*M] machine_rj_dict[(r,j)] in FILE order                 */
    const int *count;     /* [S*R] count_sr_dict[s][r]                                  */
    const int *arrive;    /* [S]   time_arrive_s_dict                                   */
    const int *delivery;  /* [S]   time_delivery_s_dict                                 */
} fjo_instance;

/* LP hook: called at every reset_object_add (class_FJSSP.py:239) with the live
 * Q[k] = len(task_unprocessed_list) and n_now[k] = len(job_now_list); must fill
 * x[k*M+m].  Returns 0 on success. */
typedef int (*fjo_lp_fn)(void *user, const int *Q, const int *n_now, double *x);

typedef struct {
    int k_sel, m_sel;       /* chosen operation type and machine                */
    int job_kind, job_n;    /* dispatched job (r, n)                            */
    int step_time;          /* self.step_time after the step                    */
    int64_t delay_time_sum; /* self.delay_time_sum after the step               */
} fjo_trace;

/* Variants share one skeleton. */
enum { FJO_SO_FJSSP = 0, FJO_SO_SFJSP = 1, FJO_MO_FJSSP_DISCRETES = 2, FJO_MO_DFJSP = 4, FJO_SO_DFJSP = 5 };

fjo_env *fjo_create(const fjo_instance *inst, int variant);
void     fjo_destroy(fjo_env *e);
void     fjo_set_lp(fjo_env *e, fjo_lp_fn fn, void *user);
/* env-internal random.choice replacement (SO_FJSSP.py:295,319): counter-based
 * stream, idx = hi32(splitmix64(seed + n_calls)) * len >> 32. */
void     fjo_set_rng(fjo_env *e, uint64_t seed);
int      fjo_state_size(const fjo_env *e);

/* reset(): SO_FJSSP.py:51-76. state has fjo_state_size() doubles. */
int fjo_reset(fjo_env *e, double *state);
/* step(): SO_FJSSP.py:168-265.  Returns 0, or <0 on MyError / stepping a done env. */
int fjo_step(fjo_env *e, int a0, int a1, double *state, double *reward, int *done, fjo_trace *tr);

/* MO_FJSSP_discretes.py:88 step(action, weight_vector, completion, tardiness);
 * completion/tardiness <= 0 stand for None. */
int fjo_step_mo(fjo_env *e, int action, double w0, double w1, double completion, double tardiness,
                double *state, double *reward, int *done, fjo_trace *tr);
/* A fixed fluid solution x[K*M] used for every LP instead of the hook (single-order instances: the one LP of
 * reset); keeps the interpreter out of timing loops. */
void fjo_set_fixed_x(fjo_env *e, const double *x);
/* MO_DFJSP(_breakdown).py: machine data of the dynamic multi-objective env (power[K*M] k-major, idle_power[M],
 * bk_n[M] breakdown windows per machine, bk[] flattened (start, end) pairs machine-major; all zero windows
 * = MO_DFJSP.py).  Must be called before fjo_reset for variant FJO_MO_DFJSP. */
void fjo_set_dynamic(fjo_env *e, const int *power, const int *idle_power, const int *bk_n, const int *bk);
/* MO_DFJSP_breakdown.py:189 step(action, reward_policy, completion, tardiness, energy_consumption) */
int fjo_step_dyn(fjo_env *e, int a0, int a1, int policy, double completion, double tardiness, double energy,
                 double *state, double *reward, int *done, fjo_trace *tr);
int64_t fjo_energy(const fjo_env *e);
/* SO_SFJSP.py:85 step(action): flat action in [0, 20). */
int fjo_step_sf(fjo_env *e, int action, double *state, double *reward, int *done, fjo_trace *tr);
/* self.DDT as the instance source parsed it (static state element 0 of the MO variant). */
void fjo_set_ddt(fjo_env *e, double ddt);

/* reset + step through actions[t][2] until done or max_T, entirely in C (cpu_baseline
 * timing without interpreter overhead).  Returns the number of steps, <0 on error. */
int fjo_play(fjo_env *e, const unsigned char *actions, int max_T, double *reward_sum);
long fjo_play_many(fjo_env **envs, int n, const unsigned char *const *actions, int max_T, int reps);

/* read-back of attributes agents/harnesses read. */
int     fjo_step_time(const fjo_env *e);
int     fjo_step_count(const fjo_env *e);
int64_t fjo_delay_time_sum(const fjo_env *e);
int     fjo_makespan(const fjo_env *e);            /* max machine.time_end       */
int     fjo_completion_time(const fjo_env *e);     /* subclasses' completion_time */
void    fjo_machine_time_end(const fjo_env *e, int *out /*[M]*/);
double  fjo_fluid_completed_time(const fjo_env *e);
/* fluid tables after the last LP (for cross-checking the device tables). */
void    fjo_fluid_tables(const fjo_env *e, double *rate /*[K*M]*/, double *arr /*[K*M]*/,
                         double *rate_sum /*[K]*/, double *time_sum /*[K]*/);

/* CPython 3.10 `list(set(a) & set(b))` order for small non-negative ints
 * (SO_FJSSP.py:302-303).  Returns the result length. */
int fjo_pyset_and_list(const int *a, int na, const int *b, int nb, int *out);

#ifdef __cplusplus
}
#endif
#endif
