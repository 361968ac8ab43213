/*
 * oracle/fjsp_oracle.c -- TEST INFRASTRUCTURE ONLY (see fjsp_oracle.h).
 *
 * Plain-C restatement of the reference environment with its Python object
 * model kept as explicit ordered lists, so list order / positional-index /
 * first-extremum tie-break semantics are the reference's own.  Every function
 * cites the reference lines it follows (paths relative to /root/reference).
 *
 * Numeric rules kept (SURVEY.md section 7 "hard parts"):
 *   - all float arithmetic is IEEE f64 in Python's evaluation order, no FMA
 *     contraction (build with -ffp-contract=off), sums are left-to-right from 0;
 *   - math.pow(x, 2) is libm pow() (NOT x*x: glibc's pow differs from x*x by
 *     1 ulp in ~0.08 % of arguments; build with -fno-builtin-pow);
 *   - round() is half-to-even on the correctly rounded quotient;
 *   - max()/min(key=) return the FIRST extremum in iteration order;
 *   - list(set(a) & set(b)) iterates in CPython 3.10 set-table order.
 */
#include "fjsp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ lists */
typedef struct { int *v; int n, cap; } ilist;

static void il_push(ilist *l, int x) {
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 8;
        l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap);
    }
    l->v[l->n++] = x;
}
/* list.remove(x): first occurrence, keeps order */
static int il_remove(ilist *l, int x) {
    for (int i = 0; i < l->n; ++i)
        if (l->v[i] == x) {
            memmove(l->v + i, l->v + i + 1, sizeof(int) * (size_t)(l->n - i - 1));
            l->n--;
            return i;
        }
    return -1;
}
static void il_clear(ilist *l) { l->n = 0; }
static void il_free(ilist *l) { free(l->v); l->v = NULL; l->n = l->cap = 0; }

/* --------------------------------------------- CPython 3.10 set emulation */
/* Objects/setobject.c (3.10): set_add_entry / set_table_resize / set_intersection.
 * Keys are small non-negative ints, hash(i) == i, no deletions. */
#define PYSET_MINSIZE 8
#define PYSET_MAXSLOTS 1024
#define LINEAR_PROBES 9
#define PERTURB_SHIFT 5
typedef struct { int key[PYSET_MAXSLOTS]; unsigned char used[PYSET_MAXSLOTS]; int mask, fill; } pyset;

static void pyset_init(pyset *s) { memset(s->used, 0, PYSET_MINSIZE); s->mask = PYSET_MINSIZE - 1; s->fill = 0; }

static void pyset_insert_clean(int *key, unsigned char *used, int mask, int k) {
    size_t perturb = (size_t)k, i = (size_t)k & (size_t)mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= (size_t)mask) ? LINEAR_PROBES : 0;
        do {
            if (!used[e]) { used[e] = 1; key[e] = k; return; }
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & (size_t)mask;
    }
}

static void pyset_resize(pyset *s, int minused) {
    int newsize = PYSET_MINSIZE;
    while (newsize <= minused) newsize <<= 1;
    int okey[PYSET_MAXSLOTS]; unsigned char oused[PYSET_MAXSLOTS];
    int osize = s->mask + 1;
    memcpy(okey, s->key, sizeof(int) * (size_t)osize);
    memcpy(oused, s->used, (size_t)osize);
    memset(s->used, 0, (size_t)newsize);
    s->mask = newsize - 1;
    for (int i = 0; i < osize; ++i)
        if (oused[i]) pyset_insert_clean(s->key, s->used, s->mask, okey[i]);
}

static void pyset_add(pyset *s, int k) {
    size_t mask = (size_t)s->mask, perturb = (size_t)k, i = (size_t)k & mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        do {
            if (!s->used[e]) goto found_unused;
            if (s->key[e] == k) return; /* already present */
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
        continue;
    found_unused:
        s->used[e] = 1; s->key[e] = k; s->fill++;
        if ((size_t)s->fill * 5 < mask * 3) return;
        pyset_resize(s, s->fill * 4); /* used == fill (no dummies); used <= 50000 */
        return;
    }
}
static int pyset_contains(const pyset *s, int k) {
    size_t mask = (size_t)s->mask, perturb = (size_t)k, i = (size_t)k & mask;
    for (;;) {
        size_t e = i;
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        do {
            if (!s->used[e]) return 0;
            if (s->key[e] == k) return 1;
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

/* list(set(a) & set(b)) -- SO_FJSSP.py:302-303 */
int fjo_pyset_and_list(const int *a, int na, const int *b, int nb, int *out) {
    pyset sa, sb, res;
    pyset_init(&sa); pyset_init(&sb); pyset_init(&res);
    for (int i = 0; i < na; ++i) pyset_add(&sa, a[i]);
    for (int i = 0; i < nb; ++i) pyset_add(&sb, b[i]);
    const pyset *so = &sa, *other = &sb;
    if (other->fill > so->fill) { const pyset *t = so; so = other; other = t; }
    for (int i = 0; i <= other->mask; ++i)
        if (other->used[i] && pyset_contains(so, other->key[i])) pyset_add(&res, other->key[i]);
    int n = 0;
    for (int i = 0; i <= res.mask; ++i)
        if (res.used[i]) out[n++] = res.key[i];
    return n;
}

/* ------------------------------------------------------------- the env */
typedef struct { int kind, n, due, time_arrive, next_j; } job_t;

struct fjo_env {
    int variant;
    /* instance */
    int R, M, K, S;
    int *Jr, *koff, *kind_of, *stage_of;
    int *p, *elig_n, *elig_list;
    int *count, *arrive, *delivery;
    /* kind_task_m_dict[m]: r-major list of k eligible on m (SO_DFJSP_instance_read.py:26) */
    ilist *ktm;
    /* LP hook / rng */
    fjo_lp_fn lp; void *lp_user;
    int class_fjsp;             /* SO_DFJSP.py: SO_FJSSP semantics over class_FJSP.py */
    double *fixed_x;            /* when set: the fluid solution of every LP (single-order instances: one LP per reset) */
    uint64_t rng_seed, rng_calls;
    /* jobs (job_dict, arrival order) */
    job_t *jobs; int njobs, jobs_cap;
    /* Kind */
    int *kind_arrived;          /* len(job_arrive_list)                        */
    ilist *kind_unproc;         /* Kind.job_unprocessed_list                   */
    /* Tasks (per k) */
    ilist *job_now;             /* job_now_list (FIFO by append order)         */
    ilist *job_unproc;          /* job_unprocessed_list == task_unprocessed_list (as jobs) */
    int *processed;             /* len(task_processed_list)                    */
    int *Q0, *fluid_number;     /* fluid_unprocessed_number_start, fluid_number */
    double *Qf;                 /* fluid_unprocessed_number                    */
    double *rate_sum, *time_sum;
    int *fl_n, *fl_list;        /* fluid_machine_list per k, [K*M]             */
    /* Machine */
    int *mstate, *tend, *mjob;
    double *rate, *arr, *un, *fu; /* [K*M] fluid_process_rate / arrival / unprocessed / fluid_unprocessed */
    double *x;
    double fluid_completed_time;
    /* orders not yet arrived */
    int next_order;
    /* update_parameter side effects (SO_FJSSP.py:36-41) */
    ilist delay_e_list, delay_a_list;
    int *delay_time_a; double *delay_time_e, *urgency; int *due_min;
    /* scalars */
    int step_count, step_time, order_arrive_time, done;
    int64_t delay_sum_last, delay_sum, delay_processed, delay_unprocessed;
    int completion_time, completion_time_last;
    double obs[16], last_obs[16];
    double static_state[8];
    int n_obs, n_static;
    /* MO_DFJSP(_breakdown): powers, breakdown windows, energy (class_MODFJSP.py:176-178, MO_DFJSP_instance_read.py:56-73) */
    int *power, *idle_power, *bk_n, *bk_off, *bk;
    int *tlast, *ntask;         /* per machine: time_end of its last task, len(task_list) */
    int64_t energy, energy_last;
    int order_count_attr;
};

static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* random.choice(list) replacement: index into a list of length n */
static int rng_choice(fjo_env *e, int n) {
    uint64_t u = splitmix64(e->rng_seed + e->rng_calls++);
    return (int)(((u >> 32) * (uint64_t)n) >> 32);
}

/* Python round(): half to even on an already correctly rounded double */
static long py_round(double v) { return (long)nearbyint(v); }

fjo_env *fjo_create(const fjo_instance *in, int variant) {
    /* environments/SO_DFJSP.py is SO_FJSSP.py over class_FJSP.py instead of class_FJSSP.py: job due date = the
     * order's delivery time (class_FJSP.py:229) and Machine.gap_ave without the 1e-18 (:159) */
    const int class_fjsp = variant == FJO_SO_DFJSP;
    if (class_fjsp) variant = FJO_SO_FJSSP;
    if (variant != FJO_SO_FJSSP && variant != FJO_SO_SFJSP && variant != FJO_MO_FJSSP_DISCRETES && variant != FJO_MO_DFJSP) return NULL;
    fjo_env *e = (fjo_env *)calloc(1, sizeof(*e));
    e->class_fjsp = class_fjsp;
    e->variant = variant;
    e->R = in->R; e->M = in->M; e->K = in->K; e->S = in->S;
    int R = e->R, M = e->M, K = e->K, S = e->S;
#define DUPI(dst, src, n) do { dst = (int *)malloc(sizeof(int) * (size_t)(n)); memcpy(dst, src, sizeof(int) * (size_t)(n)); } while (0)
    DUPI(e->Jr, in->Jr, R); DUPI(e->p, in->p, K * M); DUPI(e->elig_n, in->elig_n, K);
    DUPI(e->elig_list, in->elig_list, K * M); DUPI(e->count, in->count, S * R);
    DUPI(e->arrive, in->arrive, S); DUPI(e->delivery, in->delivery, S);
    e->koff = (int *)calloc((size_t)R + 1, sizeof(int));
    e->kind_of = (int *)calloc((size_t)K, sizeof(int));
    e->stage_of = (int *)calloc((size_t)K, sizeof(int));
    for (int r = 0; r < R; ++r) {
        e->koff[r + 1] = e->koff[r] + e->Jr[r];
        for (int j = 0; j < e->Jr[r]; ++j) { e->kind_of[e->koff[r] + j] = r; e->stage_of[e->koff[r] + j] = j; }
    }
    e->ktm = (ilist *)calloc((size_t)M, sizeof(ilist));
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k)
            if (e->p[k * M + m] > 0) il_push(&e->ktm[m], k);
    e->kind_arrived = (int *)calloc((size_t)R, sizeof(int));
    e->kind_unproc = (ilist *)calloc((size_t)R, sizeof(ilist));
    e->job_now = (ilist *)calloc((size_t)K, sizeof(ilist));
    e->job_unproc = (ilist *)calloc((size_t)K, sizeof(ilist));
    e->processed = (int *)calloc((size_t)K, sizeof(int));
    e->Q0 = (int *)calloc((size_t)K, sizeof(int));
    e->fluid_number = (int *)calloc((size_t)K, sizeof(int));
    e->Qf = (double *)calloc((size_t)K, sizeof(double));
    e->rate_sum = (double *)calloc((size_t)K, sizeof(double));
    e->time_sum = (double *)calloc((size_t)K, sizeof(double));
    e->fl_n = (int *)calloc((size_t)K, sizeof(int));
    e->fl_list = (int *)calloc((size_t)K * M, sizeof(int));
    e->mstate = (int *)calloc((size_t)M, sizeof(int));
    e->tend = (int *)calloc((size_t)M, sizeof(int));
    e->mjob = (int *)calloc((size_t)M, sizeof(int));
    e->rate = (double *)calloc((size_t)K * M, sizeof(double));
    e->arr = (double *)calloc((size_t)K * M, sizeof(double));
    e->un = (double *)calloc((size_t)K * M, sizeof(double));
    e->fu = (double *)calloc((size_t)K * M, sizeof(double));
    e->x = (double *)calloc((size_t)K * M, sizeof(double));
    e->delay_time_a = (int *)calloc((size_t)K, sizeof(int));
    e->delay_time_e = (double *)calloc((size_t)K, sizeof(double));
    e->urgency = (double *)calloc((size_t)K, sizeof(double));
    e->due_min = (int *)calloc((size_t)K, sizeof(int));
    e->n_obs = 10; e->n_static = 0;
    if (variant == FJO_MO_FJSSP_DISCRETES) { e->n_obs = 9; e->n_static = 7; }
    if (variant == FJO_SO_SFJSP) { e->n_obs = 9; e->n_static = 0; }
    if (variant == FJO_MO_DFJSP) { e->n_obs = 15; e->n_static = 0; }
    e->tlast = (int *)calloc((size_t)M, sizeof(int));
    e->ntask = (int *)calloc((size_t)M, sizeof(int));
    return e;
}

void fjo_destroy(fjo_env *e) {
    if (!e) return;
    for (int m = 0; m < e->M; ++m) il_free(&e->ktm[m]);
    for (int r = 0; r < e->R; ++r) il_free(&e->kind_unproc[r]);
    for (int k = 0; k < e->K; ++k) { il_free(&e->job_now[k]); il_free(&e->job_unproc[k]); }
    il_free(&e->delay_e_list); il_free(&e->delay_a_list);
    free(e->Jr); free(e->koff); free(e->kind_of); free(e->stage_of); free(e->p); free(e->elig_n);
    free(e->elig_list); free(e->count); free(e->arrive); free(e->delivery); free(e->ktm);
    free(e->kind_arrived); free(e->kind_unproc); free(e->job_now); free(e->job_unproc);
    free(e->processed); free(e->Q0); free(e->fluid_number); free(e->Qf); free(e->rate_sum);
    free(e->time_sum); free(e->fl_n); free(e->fl_list); free(e->mstate); free(e->tend); free(e->mjob);
    free(e->rate); free(e->arr); free(e->un); free(e->fu); free(e->x); free(e->delay_time_a);
    free(e->delay_time_e); free(e->urgency); free(e->due_min); free(e->jobs);
    free(e->fixed_x);
    free(e->power); free(e->idle_power); free(e->bk_n); free(e->bk_off); free(e->bk); free(e->tlast); free(e->ntask);
    free(e);
}

void fjo_set_lp(fjo_env *e, fjo_lp_fn fn, void *user) { e->lp = fn; e->lp_user = user; }
void fjo_set_fixed_x(fjo_env *e, const double *x) {
    free(e->fixed_x);
    e->fixed_x = (double *)malloc(sizeof(double) * (size_t)e->K * e->M);
    memcpy(e->fixed_x, x, sizeof(double) * (size_t)e->K * e->M);
}
void fjo_set_dynamic(fjo_env *e, const int *power, const int *idle_power, const int *bk_n, const int *bk) {
    int K = e->K, M = e->M, tot = 0;
    e->power = (int *)malloc(sizeof(int) * (size_t)K * M); memcpy(e->power, power, sizeof(int) * (size_t)K * M);
    e->idle_power = (int *)malloc(sizeof(int) * (size_t)M); memcpy(e->idle_power, idle_power, sizeof(int) * (size_t)M);
    e->bk_n = (int *)malloc(sizeof(int) * (size_t)M); memcpy(e->bk_n, bk_n, sizeof(int) * (size_t)M);
    e->bk_off = (int *)calloc((size_t)M + 1, sizeof(int));
    for (int m = 0; m < M; ++m) { e->bk_off[m] = tot; tot += bk_n[m]; }
    e->bk_off[M] = tot;
    e->bk = (int *)malloc(sizeof(int) * (size_t)(2 * tot + 2));
    if (tot) memcpy(e->bk, bk, sizeof(int) * (size_t)tot * 2);
}
void fjo_set_rng(fjo_env *e, uint64_t seed) { e->rng_seed = seed; e->rng_calls = 0; }
int  fjo_state_size(const fjo_env *e) { return e->n_static + 2 * e->n_obs; }

/* class_FJSSP.py:173-190 reset_parameter -- with FRESH-OBJECT semantics: the
 * reference writes machine.machine_state (:187) and so never clears
 * machine.state across reset(); every oracle episode is a new object
 * (SURVEY.md section 7), i.e. state starts 0. */
static void reset_parameter(fjo_env *e) {
    for (int r = 0; r < e->R; ++r) { e->kind_arrived[r] = 0; il_clear(&e->kind_unproc[r]); }
    for (int k = 0; k < e->K; ++k) { il_clear(&e->job_now[k]); il_clear(&e->job_unproc[k]); e->processed[k] = 0; }
    for (int m = 0; m < e->M; ++m) { e->mstate[m] = 0; e->tend[m] = 0; e->mjob[m] = -1; e->tlast[m] = 0; e->ntask[m] = 0; }
    e->njobs = 0;
}

/* class_FJSSP.py:282-306 update_fluid_parameter (+ :192-203 reset_fluid_parameter).
 * x.items() order = sorted (m, (r,j)) (the shim's variable order),
 * so each k's fluid_process_rate_m_dict / fluid_machine_list is filled in ascending m. */
static void update_fluid_parameter(fjo_env *e) {
    int K = e->K, M = e->M;
    for (int k = 0; k < K; ++k) e->fl_n[k] = 0;
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < e->ktm[m].n; ++i) {
            int k = e->ktm[m].v[i];
            double process_rate = 1.0 / (double)e->p[k * M + m];     /* class_FJSSP.py:164 */
            double rate = e->x[k * M + m];
            e->rate[k * M + m] = rate * process_rate;                 /* :288-289 */
            if (rate != 0) e->fl_list[k * M + e->fl_n[k]++] = m;      /* :290-292 */
        }
    for (int k = 0; k < K; ++k) {
        double s = 0.0;                                               /* :294 sum(dict.values()) */
        for (int m = 0; m < M; ++m)
            if (e->p[k * M + m] > 0) s = s + e->rate[k * M + m];
        e->rate_sum[k] = s;
        e->time_sum[k] = 1.0 / s;                                     /* :295 */
    }
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < e->ktm[m].n; ++i) {
            int k = e->ktm[m].v[i];
            double a = ((double)e->Q0[k] * e->rate[k * M + m]) / e->rate_sum[k]; /* :300-302 */
            e->arr[k * M + m] = a; e->un[k * M + m] = a; e->fu[k * M + m] = a;   /* :304-306 */
        }
}

/* class_FJSSP.py:205-244 reset_object_add */
static int reset_object_add(fjo_env *e, int s) {
    int R = e->R, K = e->K, M = e->M;
    for (int r = 0; r < R; ++r) {
        int n_start = e->kind_arrived[r];
        int cnt = e->count[s * R + r];
        int n_end = n_start + cnt;
        /* :214-215  round(delivery * J_r / count) */
        long r_due = py_round((double)((long)e->delivery[s] * e->Jr[r]) / (double)cnt);
        for (int n = n_start; n < n_end; ++n) {
            if (e->njobs == e->jobs_cap) {
                e->jobs_cap = e->jobs_cap ? e->jobs_cap * 2 : 64;
                e->jobs = (job_t *)realloc(e->jobs, sizeof(job_t) * (size_t)e->jobs_cap);
            }
            int id = e->njobs++;
            job_t *jb = &e->jobs[id];
            jb->kind = r; jb->n = n; jb->next_j = 0; jb->time_arrive = e->arrive[s];
            jb->due = (e->variant == FJO_MO_DFJSP || e->class_fjsp) ? e->delivery[s] /* class_MODFJSP.py:224, class_FJSP.py:229 */
                                                   : (int)py_round((double)(r_due * n) / (double)cnt);           /* :218 */
            e->kind_arrived[r]++;
            il_push(&e->kind_unproc[r], id);
            il_push(&e->job_now[e->koff[r]], id);                                  /* :225 */
            for (int j = 0; j < e->Jr[r]; ++j) il_push(&e->job_unproc[e->koff[r] + j], id); /* :230-231 */
        }
    }
    for (int k = 0; k < K; ++k) {                                                  /* :234-237 */
        e->fluid_number[k] = e->job_now[k].n;
        e->Q0[k] = e->job_unproc[k].n;
        e->Qf[k] = (double)e->Q0[k];
    }
    if (!e->lp) return -10;
    if (e->fixed_x) memcpy(e->x, e->fixed_x, sizeof(double) * (size_t)e->K * e->M);  /* x supplied once (static instance) */
    else if (e->lp(e->lp_user, e->Q0, e->fluid_number, e->x) != 0) return -11;    /* :239 fluid_model */
    /* :276-278 fluid_completed_time = max Q / rate_sum, rate_sum summed over machine_rj_dict order */
    {
        double best = 0.0; int first = 1;
        for (int k = 0; k < K; ++k) {
            double s2 = 0.0;
            for (int i = 0; i < e->elig_n[k]; ++i) {
                int m = e->elig_list[k * M + i];
                s2 = s2 + e->x[k * M + m] * (1.0 / (double)e->p[k * M + m]);
            }
            double v = (double)e->Q0[k] / s2;
            if (first || v > best) { best = v; first = 0; }
        }
        e->fluid_completed_time = best;
    }
    update_fluid_parameter(e);
    return 0;
}

/* ------------------------------------------------------ availability (a8) */
static int machine_idle_list(const fjo_env *e, int *out) {          /* SO_FJSSP.py:369-371 */
    int n = 0;
    for (int m = 0; m < e->M; ++m) if (e->mstate[m] == 0) out[n++] = m;
    return n;
}
static int k_available(const fjo_env *e, int k, int fluid) {        /* :373-381 */
    if (e->job_now[k].n == 0) return 0;
    if (!fluid) {
        for (int i = 0; i < e->elig_n[k]; ++i) if (e->mstate[e->elig_list[k * e->M + i]] == 0) return 1;
    } else {
        for (int i = 0; i < e->fl_n[k]; ++i) if (e->mstate[e->fl_list[k * e->M + i]] == 0) return 1;
    }
    return 0;
}
static int available_list(const fjo_env *e, int fluid, int *out) {
    int n = 0;
    for (int k = 0; k < e->K; ++k) if (k_available(e, k, fluid)) out[n++] = k;
    return n;
}

/* class_FJSSP.py:66-84 */
static double tasks_gap(const fjo_env *e, int k) { return (double)e->job_unproc[k].n - e->Qf[k]; }
static double tasks_gap_rate(const fjo_env *e, int k) { return ((double)e->job_unproc[k].n - e->Qf[k]) / (double)e->Q0[k]; }
static double tasks_finish_rate(const fjo_env *e, int k) {
    return (double)e->processed[k] / (double)(e->job_unproc[k].n + e->processed[k]);
}
static int tasks_due_date_min(const fjo_env *e, int k) {
    int best = e->jobs[e->job_now[k].v[0]].due;
    for (int i = 1; i < e->job_now[k].n; ++i) { int d = e->jobs[e->job_now[k].v[i]].due; if (d < best) best = d; }
    return best;
}
/* class_FJSSP.py:137-146 */
static double machine_gap_rj(const fjo_env *e, int m, int k) { return e->un[k * e->M + m] - e->fu[k * e->M + m]; }
static double machine_gap_ave(const fjo_env *e, int m) {
    double s = 0.0;
    for (int i = 0; i < e->ktm[m].n; ++i) s = s + machine_gap_rj(e, m, e->ktm[m].v[i]);
    if (e->variant == FJO_MO_DFJSP || e->class_fjsp) return s / (double)e->ktm[m].n;   /* class_MODFJSP.py:158-159, class_FJSP.py:159 */
    return s / ((double)e->ktm[m].n + 1e-18);
}

/* SO_FJSSP.py:99-166 update_parameter */
static void update_parameter(fjo_env *e, double *dro_a, double *dro_e, double *drj_a, double *drj_e) {
    long delay_task_number_a = 0, delay_task_number_e = 0, task_number = 0;
    long delay_job_number_a = 0, delay_job_number_e = 0, job_number = 0;
    e->delay_unprocessed = 0;
    il_clear(&e->delay_e_list); il_clear(&e->delay_a_list);
    int t = e->step_time;
    for (int r = 0; r < e->R; ++r) {                                            /* :116-124 */
        job_number += e->kind_unproc[r].n;
        int kend = e->koff[r] + e->Jr[r] - 1;
        for (int idx = 0; idx < e->job_unproc[kend].n; ++idx) {
            const job_t *jb = &e->jobs[e->job_unproc[kend].v[idx]];
            if (t > jb->due) { delay_job_number_a++; e->delay_unprocessed += (t - jb->due); }
            if ((double)t + e->time_sum[kend] * (double)(idx + 1) > (double)jb->due) delay_job_number_e++;
        }
    }
    for (int k = 0; k < e->K; ++k) {                                            /* :126-154 */
        int residue = e->job_unproc[k].n;
        task_number += residue;
        int cnt_a = 0, cnt_e = 0;
        int max_a = 0; double max_e = 0.0, sum_e = 0.0;
        for (int idx = 0; idx < residue; ++idx) {
            const job_t *jb = &e->jobs[e->job_unproc[k].v[idx]];
            if (t > jb->due) cnt_a++;
            double est = (double)t + e->time_sum[k] * (double)(idx + 1);
            if (est > (double)jb->due) cnt_e++;
            int da = t - jb->due;
            double de = est - (double)jb->due;
            if (idx == 0 || da > max_a) max_a = da;
            if (idx == 0 || de > max_e) max_e = de;
            sum_e = sum_e + de;
        }
        delay_task_number_a += cnt_a; delay_task_number_e += cnt_e;
        if (k_available(e, k, 0)) {                                             /* :145 */
            if (cnt_a > 0) { il_push(&e->delay_a_list, k); e->delay_time_a[k] = max_a; }
            if (cnt_e > 0) { il_push(&e->delay_e_list, k); e->delay_time_e[k] = max_e; }
            e->urgency[k] = sum_e / (double)residue;                            /* :153 */
            e->due_min[k] = tasks_due_date_min(e, k);                           /* :154 */
        }
    }
    if (!e->done) {                                                             /* :156-165 */
        *dro_a = (double)delay_task_number_a / (double)task_number;
        *dro_e = (double)delay_task_number_e / (double)task_number;
        *drj_a = (double)delay_job_number_a / (double)job_number;
        *drj_e = (double)delay_job_number_e / (double)job_number;
    } else { *dro_a = *dro_e = *drj_a = *drj_e = 0.0; }
}

static double pop_std_k(const fjo_env *e, double (*f)(const fjo_env *, int), double ave) {
    double s = 0.0;
    for (int k = 0; k < e->K; ++k) s = s + pow(f(e, k) - ave, 2.0);
    return sqrt(s / (double)e->K);
}
static double mean_k(const fjo_env *e, double (*f)(const fjo_env *, int)) {
    double s = 0.0;
    for (int k = 0; k < e->K; ++k) s = s + f(e, k);
    return s / (double)e->K;
}

/* SO_FJSSP.py:78-97 / MO_FJSSP_discretes.py:66-86 state_extract */
static void state_extract(fjo_env *e, double *o) {
    int M = e->M;
    long tsum = 0;
    for (int m = 0; m < M; ++m) tsum += e->tend[m];
    double ct_m_ave = (double)tsum / (double)M;                                 /* :384-385 */
    double s = 0.0;
    for (int m = 0; m < M; ++m) s = s + pow((double)e->tend[m] - ct_m_ave, 2.0);
    double ct_m_std = sqrt(s / (double)M);
    if (e->variant == FJO_MO_DFJSP) {                                           /* MO_DFJSP_breakdown.py:94-118 */
        int tmp[4096];
        double ratio_idle = (double)available_list(e, 1, tmp) / ((double)available_list(e, 0, tmp) + 1e-08);
        double cro_a = mean_k(e, tasks_finish_rate), cro_s = pop_std_k(e, tasks_finish_rate, cro_a);
        double gap_a = mean_k(e, tasks_gap_rate), gap_s = pop_std_k(e, tasks_gap_rate, gap_a);
        double gs = 0.0;
        for (int m = 0; m < M; ++m) gs = gs + machine_gap_ave(e, m);
        double gap_m_ave = gs / (double)M;
        double g2 = 0.0;
        for (int m = 0; m < M; ++m) g2 = g2 + pow(machine_gap_ave(e, m) - gap_m_ave, 2.0);
        double gap_m_std = sqrt(g2 / (double)M);
        double da, de, ja, je;
        update_parameter(e, &da, &de, &ja, &je);
        o[0] = e->static_state[0]; o[1] = (double)M; o[2] = (double)e->S; o[3] = ct_m_std; o[4] = ratio_idle;
        o[5] = cro_a; o[6] = cro_s; o[7] = gap_a; o[8] = gap_s; o[9] = gap_m_ave; o[10] = gap_m_std;
        o[11] = da; o[12] = de; o[13] = ja; o[14] = je;
        return;
    }
    if (e->variant == FJO_SO_SFJSP) {                                           /* SO_SFJSP.py:64-83 */
        int tmp[4096];
        double M_idle_ratio = (double)machine_idle_list(e, tmp) / (double)M;
        double ratio_idle = (double)available_list(e, 1, tmp) / ((double)available_list(e, 0, tmp) + 1e-08);
        double cro_a = mean_k(e, tasks_finish_rate), cro_s = pop_std_k(e, tasks_finish_rate, cro_a);
        double gap_a = mean_k(e, tasks_gap_rate), gap_s = pop_std_k(e, tasks_gap_rate, gap_a);
        double gs = 0.0;
        for (int m = 0; m < M; ++m) gs = gs + machine_gap_ave(e, m);
        double gap_m_ave = gs / (double)M;
        double g2 = 0.0;
        for (int m = 0; m < M; ++m) g2 = g2 + pow(machine_gap_ave(e, m) - gap_m_ave, 2.0);
        double gap_m_std = sqrt(g2 / (double)M);
        o[0] = M_idle_ratio; o[1] = ct_m_std; o[2] = cro_a; o[3] = cro_s; o[4] = ratio_idle;
        o[5] = gap_a; o[6] = gap_s; o[7] = gap_m_ave; o[8] = gap_m_std;
        return;                               /* no update_parameter() call in this subclass */
    }
    double cro_ave = mean_k(e, tasks_finish_rate), cro_std = pop_std_k(e, tasks_finish_rate, cro_ave);
    double gap_ave = mean_k(e, tasks_gap_rate), gap_std = pop_std_k(e, tasks_gap_rate, gap_ave);
    double dro_a, dro_e, drj_a, drj_e;
    update_parameter(e, &dro_a, &dro_e, &drj_a, &drj_e);
    int i = 0;
    if (e->variant == FJO_SO_FJSSP) o[i++] = (double)M;
    o[i++] = ct_m_std; o[i++] = cro_ave; o[i++] = cro_std; o[i++] = gap_ave; o[i++] = gap_std;
    o[i++] = dro_a; o[i++] = dro_e; o[i++] = drj_a; o[i++] = drj_e;
}

static void compose_state(const fjo_env *e, double *state) {
    int i = 0;
    for (int a = 0; a < e->n_static; ++a) state[i++] = e->static_state[a];
    for (int a = 0; a < e->n_obs; ++a) state[i++] = e->obs[a];
    for (int a = 0; a < e->n_obs; ++a) state[i++] = e->obs[a] - e->last_obs[a];
}

/* MO_FJSSP_discretes.py:55-64 static_state_extract (DDT supplied by fjo_set_ddt) */
static void static_state_extract(fjo_env *e, double ddt) {
    int R = e->R;
    long ns = 0, js = 0;
    for (int r = 0; r < R; ++r) { ns += e->count[r]; js += e->Jr[r]; }
    double N_ave = (double)ns / (double)R, J_ave = (double)js / (double)R;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < R; ++r) a = a + pow((double)e->count[r] - N_ave, 2.0);
    for (int r = 0; r < R; ++r) b = b + pow((double)e->Jr[r] - J_ave, 2.0);
    double N_std = sqrt(a / (double)R), J_std = sqrt(b / (double)R);
    e->static_state[0] = ddt; e->static_state[1] = (double)e->M; e->static_state[2] = (double)R;
    e->static_state[3] = N_ave; e->static_state[4] = N_std; e->static_state[5] = J_ave; e->static_state[6] = J_std;
}

/* SO_FJSSP.py:51-76 reset */
int fjo_reset(fjo_env *e, double *state) {
    e->next_order = 0;
    reset_parameter(e);
    int rc = reset_object_add(e, e->next_order++);
    if (rc) return rc;
    e->delay_sum_last = e->delay_sum = e->delay_processed = e->delay_unprocessed = 0;
    e->completion_time = e->completion_time_last = 0;
    e->energy = e->energy_last = 0;
    e->step_count = 0; e->step_time = 0; e->order_arrive_time = 0; e->done = 0;
    if (e->variant == FJO_MO_FJSSP_DISCRETES) static_state_extract(e, e->static_state[0]);
    state_extract(e, e->last_obs);
    state_extract(e, e->obs);
    compose_state(e, state);
    return 0;
}

/* SO_SFJSP.py:234-244 time_min_rj / time_min_fluid_rj: min processing time of k over the idle
 * machines of its (fluid) machine list, first minimum in list(set & set) order */
static int time_min_rj(fjo_env *e, int k, int fluid) {
    int idle[1024], sel[1024];
    int nidle = machine_idle_list(e, idle);
    int n = fluid ? fjo_pyset_and_list(idle, nidle, &e->fl_list[k * e->M], e->fl_n[k], sel)
                  : fjo_pyset_and_list(idle, nidle, &e->elig_list[k * e->M], e->elig_n[k], sel);
    int best = e->p[k * e->M + sel[0]];
    for (int i = 1; i < n; ++i) if (e->p[k * e->M + sel[i]] < best) best = e->p[k * e->M + sel[i]];
    return best;
}

/* MO_DFJSP_breakdown.py:498-508 energy_min_rj / energy_min_fluid_rj */
static long energy_min_rj(fjo_env *e, int k, int fluid) {
    int idle[1024], sel[1024];
    int nidle = machine_idle_list(e, idle);
    int n = fluid ? fjo_pyset_and_list(idle, nidle, &e->fl_list[k * e->M], e->fl_n[k], sel)
                  : fjo_pyset_and_list(idle, nidle, &e->elig_list[k * e->M], e->elig_n[k], sel);
    long best = (long)e->power[k * e->M + sel[0]] * e->p[k * e->M + sel[0]];
    for (int i = 1; i < n; ++i) { long v = (long)e->power[k * e->M + sel[i]] * e->p[k * e->M + sel[i]]; if (v < best) best = v; }
    return best;
}

/* SO_FJSSP.py:267-298 task_select.  Returns k or <0 (MyError). */
static int task_select(fjo_env *e, int rule) {
    int av[4096], fav[4096];
    int nav = available_list(e, 0, av);
    if (nav == 0) return -3;
    if (e->variant == FJO_SO_SFJSP) {                                           /* SO_SFJSP.py:169-188 */
        int nf = available_list(e, 1, fav);
        switch (rule) {
        case 1: {
            const int *l = nf ? fav : av; int n = nf ? nf : nav;
            int b = l[0]; double bv = tasks_gap(e, b);
            for (int i = 1; i < n; ++i) { double v = tasks_gap(e, l[i]); if (v > bv) { bv = v; b = l[i]; } }
            return b; }
        case 2: case 3: {
            int fluid = (rule == 2 && nf > 0);
            const int *l = fluid ? fav : av; int n = fluid ? nf : nav;
            int b = -1, bv = 0;
            for (int i = 0; i < n; ++i) { int v = time_min_rj(e, l[i], fluid); if (b < 0 || v < bv) { bv = v; b = l[i]; } }
            return b; }
        case 4: return av[rng_choice(e, nav)];
        default: return -1;
        }
    }
    if (e->variant == FJO_MO_DFJSP && rule >= 6) {                              /* MO_DFJSP_breakdown.py:357-381 */
        int nf = available_list(e, 1, fav);
        switch (rule) {
        case 6: { int b = av[0]; for (int i = 1; i < nav; ++i) if (e->due_min[av[i]] < e->due_min[b]) b = av[i]; return b; }
        case 7: case 8: {
            int fluid = (rule == 7 && nf > 0);
            const int *l = fluid ? fav : av; int n = fluid ? nf : nav;
            int b = -1; long bv = 0;
            for (int i = 0; i < n; ++i) { long v = energy_min_rj(e, l[i], fluid); if (b < 0 || v < bv) { bv = v; b = l[i]; } }
            return b; }
        case 9: case 10: {
            int fluid = (rule == 9 && nf > 0);
            const int *l = fluid ? fav : av; int n = fluid ? nf : nav;
            int b = -1, bv = 0;
            for (int i = 0; i < n; ++i) { int v = time_min_rj(e, l[i], fluid); if (b < 0 || v < bv) { bv = v; b = l[i]; } }
            return b; }
        case 11: return nf ? fav[rng_choice(e, nf)] : av[rng_choice(e, nav)];
        case 12: return av[rng_choice(e, nav)];
        default: return -1;
        }
    }
#define ARGMAX_D(list, n, key) ({ int _b = (list)[0]; double _bv = (key)[_b]; \
        for (int _i = 1; _i < (n); ++_i) { int _c = (list)[_i]; if ((key)[_c] > _bv) { _bv = (key)[_c]; _b = _c; } } _b; })
    switch (rule) {
    case 1:
        if (e->delay_e_list.n == 0) return ARGMAX_D(av, nav, e->urgency);
        return ARGMAX_D(e->delay_e_list.v, e->delay_e_list.n, e->delay_time_e);
    case 2:
        if (e->delay_a_list.n == 0) return ARGMAX_D(av, nav, e->urgency);
        { int b = e->delay_a_list.v[0];
          for (int i = 1; i < e->delay_a_list.n; ++i) { int c = e->delay_a_list.v[i]; if (e->delay_time_a[c] > e->delay_time_a[b]) b = c; }
          return b; }
    case 3: {
        int nf = available_list(e, 1, fav);
        const int *l = nf ? fav : av; int n = nf ? nf : nav;
        int b = l[0]; double bv = tasks_gap(e, b);
        for (int i = 1; i < n; ++i) { double v = tasks_gap(e, l[i]); if (v > bv) { bv = v; b = l[i]; } }
        return b; }
    case 4: {
        int nf = available_list(e, 1, fav);
        if (nf == 0) return ARGMAX_D(av, nav, e->urgency);
        return ARGMAX_D(fav, nf, e->urgency); }
    case 5: {
        int nf = available_list(e, 1, fav);
        const int *l = nf ? fav : av; int n = nf ? nf : nav;
        int b = l[0];
        for (int i = 1; i < n; ++i) if (e->due_min[l[i]] < e->due_min[b]) b = l[i];
        return b; }
    case 6:
        return av[rng_choice(e, nav)];
    default:
        return -1; /* MyError */
    }
}

/* SO_FJSSP.py:300-322 machine_select (5 rules),
 * MO_FJSSP_discretes.py:209-230 (3 rules). */
static int machine_select(fjo_env *e, int rule, int k) {
    int idle[1024], sel[1024], fsel[1024];
    int M = e->M;
    int nidle = machine_idle_list(e, idle);
    int nsel = fjo_pyset_and_list(idle, nidle, &e->elig_list[k * M], e->elig_n[k], sel);
    int nfs = fjo_pyset_and_list(idle, nidle, &e->fl_list[k * M], e->fl_n[k], fsel);
    if (nsel == 0) return -3;
#define ARGMAX_GAP(list, n) ({ int _b = (list)[0]; double _bv = machine_gap_rj(e, _b, k); \
        for (int _i = 1; _i < (n); ++_i) { double _v = machine_gap_rj(e, (list)[_i], k); if (_v > _bv) { _bv = _v; _b = (list)[_i]; } } _b; })
#define ARGMAX_GAVE(list, n) ({ int _b = (list)[0]; double _bv = machine_gap_ave(e, _b); \
        for (int _i = 1; _i < (n); ++_i) { double _v = machine_gap_ave(e, (list)[_i]); if (_v > _bv) { _bv = _v; _b = (list)[_i]; } } _b; })
#define ARGMIN_P(list, n) ({ int _b = (list)[0]; \
        for (int _i = 1; _i < (n); ++_i) { if (e->p[k * M + (list)[_i]] < e->p[k * M + _b]) _b = (list)[_i]; } _b; })
    if (e->variant == FJO_MO_FJSSP_DISCRETES) {
        switch (rule) {
        case 1: return nfs == 0 ? ARGMIN_P(sel, nsel) : ARGMAX_GAP(fsel, nfs);
        case 2: return nfs == 0 ? ARGMAX_GAVE(sel, nsel) : ARGMAX_GAVE(fsel, nfs);
        case 3: return nfs == 0 ? ARGMAX_GAP(sel, nsel) : ARGMAX_GAP(fsel, nfs);
        default: return -2;
        }
    }
    if (e->variant == FJO_MO_DFJSP) {                                           /* MO_DFJSP_breakdown.py:384-428 */
#define ARGMIN_E(list, n) ({ int _b = (list)[0]; \
        for (int _i = 1; _i < (n); ++_i) { if ((long)e->power[k * M + (list)[_i]] * e->p[k * M + (list)[_i]] < (long)e->power[k * M + _b] * e->p[k * M + _b]) _b = (list)[_i]; } _b; })
#define ARGMIN_IDLE(list, n) ({ int _b = (list)[0]; \
        for (int _i = 1; _i < (n); ++_i) { if (e->idle_power[(list)[_i]] < e->idle_power[_b]) _b = (list)[_i]; } _b; })
        switch (rule) {
        case 1: return nfs == 0 ? ARGMAX_GAP(sel, nsel) : ARGMAX_GAP(fsel, nfs);
        case 2: return nfs == 0 ? ARGMIN_P(sel, nsel) : ARGMIN_P(fsel, nfs);
        case 3: return ARGMIN_P(sel, nsel);
        case 4: return nfs == 0 ? ARGMAX_GAVE(sel, nsel) : ARGMAX_GAVE(fsel, nfs);
        case 5: return nfs == 0 ? ARGMIN_E(sel, nsel) : ARGMIN_E(fsel, nfs);
        case 6: return ARGMIN_E(sel, nsel);
        case 7: return nfs == 0 ? ARGMIN_IDLE(sel, nsel) : ARGMIN_IDLE(fsel, nfs);
        case 8: return ARGMIN_IDLE(sel, nsel);
        case 9: return nfs == 0 ? sel[rng_choice(e, nsel)] : fsel[rng_choice(e, nfs)];
        case 10: return sel[rng_choice(e, nsel)];
        default: return -2;
        }
    }
    if (e->variant == FJO_SO_SFJSP) {                                           /* SO_SFJSP.py:190-214 */
        switch (rule) {
        case 1: return nfs == 0 ? ARGMAX_GAP(sel, nsel) : ARGMAX_GAP(fsel, nfs);
        case 2: return nfs == 0 ? ARGMIN_P(sel, nsel) : ARGMIN_P(fsel, nfs);
        case 3: return ARGMIN_P(sel, nsel);
        case 4: return nfs == 0 ? ARGMAX_GAVE(sel, nsel) : ARGMAX_GAVE(fsel, nfs);
        case 5: return sel[rng_choice(e, nsel)];
        default: return -2;
        }
    }
    switch (rule) {
    case 1: return nfs == 0 ? ARGMAX_GAP(sel, nsel) : ARGMAX_GAP(fsel, nfs);
    case 2: return ARGMAX_GAP(sel, nsel);
    case 3: return ARGMIN_P(sel, nsel);
    case 4: return nfs == 0 ? ARGMAX_GAVE(sel, nsel) : ARGMAX_GAVE(fsel, nfs);
    case 5: return sel[rng_choice(e, nsel)];
    default: return -2; /* MyError */
    }
}

static long unfinished_jobs(const fjo_env *e) {
    long n = 0;
    for (int r = 0; r < e->R; ++r) n += e->kind_unproc[r].n;
    return n;
}

/* SO_FJSSP.py:168-258 / MO_FJSSP_discretes.py:88-161: everything in step() up to the reward */
static int step_core(fjo_env *e, int task_rule, int machine_rule, double *state, fjo_trace *tr) {
    if (e->done) return -4;
    int M = e->M;
    int k = task_select(e, task_rule);
    if (k < 0) return k;
    int m = machine_select(e, machine_rule, k);
    if (m < 0) return m;
    int job = e->job_now[k].v[0];                                              /* :176 */
    job_t *jb = &e->jobs[job];
    int time_end = e->step_time + e->p[k * M + m];                             /* :184 */
    int machine_end = time_end;
    if (e->variant == FJO_MO_DFJSP) {                                          /* MO_DFJSP_breakdown.py:204-231 */
        int cur = e->step_time;
        for (int q = e->bk_off[m]; q < e->bk_off[m + 1]; ++q) {
            int bs = e->bk[2 * q], be = e->bk[2 * q + 1];
            if (bs <= cur && cur < be) { int d = be - cur; time_end += d; machine_end = time_end; }
            else if (cur < bs && bs < time_end) { int d = be - bs; time_end += d; machine_end = time_end; }
            else if (bs == time_end) machine_end += (be - bs);
            else if (bs > time_end) break;
        }
    }
    jb->next_j++;                                                              /* :186-187 */
    il_remove(&e->job_now[k], job);                                            /* :189 */
    il_remove(&e->job_unproc[k], job);                                         /* :190-191 */
    e->processed[k]++;                                                         /* :192 */
    e->mstate[m] = 1; e->tend[m] = machine_end; e->mjob[m] = job;              /* :194-197 (machine_end_time for the dynamic env) */
    if (e->variant == FJO_MO_DFJSP) {                                          /* MO_DFJSP_breakdown.py:253-256 */
        e->energy += (int64_t)e->power[k * M + m] * e->p[k * M + m];
        e->ntask[m]++;
        if (e->ntask[m] >= 2) e->energy += (int64_t)(e->step_time - e->tlast[m]) * e->idle_power[m];
        e->tlast[m] = time_end;
    }
    e->un[k * M + m] -= 1;                                                     /* :198 */
    if (time_end > e->completion_time) e->completion_time = time_end;          /* MO_FJSSP_discretes.py:122 */
    if (jb->next_j == e->Jr[jb->kind]) {                                       /* :200-202 */
        il_remove(&e->kind_unproc[jb->kind], job);
        int late = time_end - jb->due;
        e->delay_processed += late > 0 ? late : 0;
    }
    if (tr) { tr->k_sel = k; tr->m_sel = m; tr->job_kind = jb->kind; tr->job_n = jb->n; }
    int tmp[4096];
    while (available_list(e, 0, tmp) == 0) {                                   /* :204 */
        int tmin = 0, have = 0;
        for (int mm = 0; mm < M; ++mm)
            if (e->tend[mm] > e->step_time && (!have || e->tend[mm] < tmin)) { tmin = e->tend[mm]; have = 1; }
        if (!have) return -5; /* reference: ValueError min() of empty sequence */
        e->step_time = tmin;                                                   /* :205-207 */
        for (int mm = 0; mm < M; ++mm)                                         /* :209-215 */
            if (e->tend[mm] == e->step_time) {
                int j2 = e->mjob[mm];
                job_t *b2 = &e->jobs[j2];
                if (b2->next_j < e->Jr[b2->kind]) il_push(&e->job_now[e->koff[b2->kind] + b2->next_j], j2);
            }
        if (e->variant == FJO_SO_FJSSP || e->variant == FJO_MO_DFJSP) {
            if (e->next_order < e->S && e->arrive[e->next_order] <= e->step_time) {       /* :218-223 */
                int s = e->next_order++;
                int rc = reset_object_add(e, s); if (rc) return rc;
                e->order_arrive_time = e->arrive[s];
            } else if (e->next_order < e->S && unfinished_jobs(e) == 0) {                 /* :224-231 */
                int s = e->next_order++;
                int rc = reset_object_add(e, s); if (rc) return rc;
                e->order_arrive_time = e->arrive[s];
                e->step_time = e->order_arrive_time;
            }
        }
        for (int mm = 0; mm < M; ++mm) if (e->tend[mm] <= e->step_time) e->mstate[mm] = 0; /* :233-235 */
        int gap_time = e->step_time - e->order_arrive_time;                    /* :237 */
        for (int kk = 0; kk < e->K; ++kk)                                      /* :238-240 */
            e->Qf[kk] = (double)e->Q0[kk] - e->rate_sum[kk] * (double)gap_time;
        for (int mm = 0; mm < M; ++mm)                                         /* :241-245 */
            for (int i = 0; i < e->ktm[mm].n; ++i) {
                int kk = e->ktm[mm].v[i];
                e->fu[kk * M + mm] = e->arr[kk * M + mm] - (double)gap_time * e->rate[kk * M + mm];
            }
        int orders_left = (e->variant == FJO_SO_FJSSP || e->variant == FJO_MO_DFJSP) ? (e->S - e->next_order) : 0;
        if (orders_left == 0 && unfinished_jobs(e) == 0) { e->done = 1; break; }           /* :247-250 */
    }
    e->step_count++;                                                           /* :252 */
    memcpy(e->last_obs, e->obs, sizeof(e->obs));
    state_extract(e, e->obs);                                                  /* :256 */
    compose_state(e, state);
    e->delay_sum = e->delay_processed + e->delay_unprocessed;                  /* :259 */
    if (tr) { tr->step_time = e->step_time; tr->delay_time_sum = e->delay_sum; }
    return 0;
}

/* SO_FJSSP.py:168-265 step; reward = compute_reward() branch 1 (:326-328) */
int fjo_step(fjo_env *e, int a0, int a1, double *state, double *reward, int *done, fjo_trace *tr) {
    int rc = step_core(e, a0 + 1, a1 + 1, state, tr);
    if (rc) return rc;
    *reward = (double)(-(e->delay_sum - e->delay_sum_last));                   /* :328, exact int */
    e->delay_sum_last = e->delay_sum;                                          /* :263 */
    e->completion_time_last = e->completion_time;
    *done = e->done;
    return 0;
}

/* MO_FJSSP_discretes.py:88-174 step(action, weight_vector, completion, tardiness);
 * actions table :26 = (task_rule in range(6)) x (machine_rule in range(3));
 * compute_reward :232-244.  completion/tardiness <= 0 stands for None. */
int fjo_step_mo(fjo_env *e, int action, double w0, double w1, double completion, double tardiness,
                double *state, double *reward, int *done, fjo_trace *tr) {
    if (e->variant != FJO_MO_FJSSP_DISCRETES) return -6;
    if (action < 0 || action >= 18) return -1;
    int rc = step_core(e, action / 3 + 1, action % 3 + 1, state, tr);
    if (rc) return rc;
    double dc = (double)(e->completion_time_last - e->completion_time);
    double dt = (double)(e->delay_sum_last - e->delay_sum);
    if (completion > 0 && tardiness > 0) *reward = dc / completion * w0 + dt / tardiness * w1;
    else if (w1 == 1) *reward = dt;
    else if (w0 == 1) *reward = dc;
    else return -7; /* MyError */
    e->delay_sum_last = e->delay_sum;
    e->completion_time_last = e->completion_time;
    *done = e->done;
    return 0;
}
void fjo_set_ddt(fjo_env *e, double ddt) { e->static_state[0] = ddt; }

/* MO_DFJSP_breakdown.py:189-328 step(action, reward_policy, completion, tardiness, energy_consumption);
 * compute_reward :430-447. */
int fjo_step_dyn(fjo_env *e, int a0, int a1, int policy, double completion, double tardiness, double energy,
                 double *state, double *reward, int *done, fjo_trace *tr) {
    if (e->variant != FJO_MO_DFJSP) return -6;
    int rc = step_core(e, a0 + 1, a1 + 1, state, tr);
    if (rc) return rc;
    double dc = (double)(e->completion_time_last - e->completion_time);
    double dt = (double)(e->delay_sum_last - e->delay_sum);
    double de = (double)(e->energy_last - e->energy);
    if (policy == 0) *reward = dc;
    else if (policy == 1) *reward = dt;
    else if (policy == 2) *reward = de;
    else if (policy == 3) *reward = tardiness > 0 ? dc / completion + dt / tardiness + de / energy : dc / completion + de / energy;
    else return -7;
    e->delay_sum_last = e->delay_sum;
    e->completion_time_last = e->completion_time;
    e->energy_last = e->energy;
    *done = e->done;
    return 0;
}
int64_t fjo_energy(const fjo_env *e) { return e->energy; }

/* SO_SFJSP.py:85-167 step(action): actions table :25 = (task_rule in range(4)) x (machine_rule in range(5));
 * reward :216-222 = -(completion_time - completion_time_last) / fluid_completed_time.
 * state_extract of this subclass never calls update_parameter, so delay_time_sum counts finished jobs only. */
int fjo_step_sf(fjo_env *e, int action, double *state, double *reward, int *done, fjo_trace *tr) {
    if (e->variant != FJO_SO_SFJSP) return -6;
    if (action < 0 || action >= 20) return -1;
    int rc = step_core(e, action / 5 + 1, action % 5 + 1, state, tr);
    if (rc) return rc;
    *reward = (double)(-(e->completion_time - e->completion_time_last)) / e->fluid_completed_time;
    e->delay_sum_last = e->delay_sum;
    e->completion_time_last = e->completion_time;
    *done = e->done;
    return 0;
}


/* reset + play actions[t][2] until done (or max_T), all in C: the cpu_baseline timing loop */
int fjo_play(fjo_env *e, const unsigned char *actions, int max_T, double *reward_sum) {
    double st[64], r, acc = 0.0;
    int done = 0, t = 0;
    int rc = fjo_reset(e, st);
    if (rc) return rc;
    while (!done && t < max_T) {
        rc = fjo_step(e, actions[2 * t], actions[2 * t + 1], st, &r, &done, NULL);
        if (rc) return rc;
        acc += r;
        t++;
    }
    if (reward_sum) *reward_sum = acc;
    return t;
}

/* timing helper: `reps` full episodes of each of n environments in one call (the caller's thread keeps
 * running without going back to the interpreter between episodes); returns the number of steps played */
long fjo_play_many(fjo_env **envs, int n, const unsigned char *const *actions, int max_T, int reps) {
    long steps = 0;
    for (int q = 0; q < reps; ++q)
        for (int i = 0; i < n; ++i) {
            int t = fjo_play(envs[i], actions[i], max_T, NULL);
            if (t < 0) return t;
            steps += t;
        }
    return steps;
}

int     fjo_step_time(const fjo_env *e) { return e->step_time; }
int     fjo_step_count(const fjo_env *e) { return e->step_count; }
int64_t fjo_delay_time_sum(const fjo_env *e) { return e->delay_sum; }
int     fjo_completion_time(const fjo_env *e) { return e->completion_time; }
int     fjo_makespan(const fjo_env *e) {
    int mx = 0;
    for (int m = 0; m < e->M; ++m) if (e->tend[m] > mx) mx = e->tend[m];
    return mx;
}
void fjo_machine_time_end(const fjo_env *e, int *out) { memcpy(out, e->tend, sizeof(int) * (size_t)e->M); }
double fjo_fluid_completed_time(const fjo_env *e) { return e->fluid_completed_time; }
void fjo_fluid_tables(const fjo_env *e, double *rate, double *arr, double *rate_sum, double *time_sum) {
    size_t km = (size_t)e->K * (size_t)e->M;
    memcpy(rate, e->rate, sizeof(double) * km); memcpy(arr, e->arr, sizeof(double) * km);
    memcpy(rate_sum, e->rate_sum, sizeof(double) * (size_t)e->K);
    memcpy(time_sum, e->time_sum, sizeof(double) * (size_t)e->K);
}
