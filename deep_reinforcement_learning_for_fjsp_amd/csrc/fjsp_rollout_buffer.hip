// On-policy rollout buffer in HBM behind the C ABI (include/fjsp_amd.h).
//
// Replaces the reference's Replay_Buffer (agents/MPPPO/Buffer.py:7-58: a Python
// deque of namedtuples that is vstack-ed and copied host->device once per
// episode) by a [T][N][...] f32 slab the batched step writes into directly and
// the PyTorch side wraps without copy, plus the discounted-return reverse scan
// of agents/MPPPO/MPPPO.py:301-312.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fjsp_amd.h"
#include "fjsp_host.h"
#include "fjsp_policy.h"

#pragma clang fp contract(off)

namespace {

// one thread per (env, state element): Buffer.py:41-45 `.float()` conversions
__global__ void append_kernel(int N, int S, const double *state, const uint8_t *actions, const double *reward,
                              const double *next_state, const uint8_t *done, const uint8_t *active, float *o_state,
                              float *o_actions, float *o_reward, float *o_next, float *o_done, float *o_valid) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * S) return;
    const size_t env = idx / S, i = idx % S;
    o_state[idx] = (float)state[idx];
    o_next[idx] = (float)next_state[idx];
    if (i == 0) {
        o_actions[env * 2] = (float)actions[env * 2];
        o_actions[env * 2 + 1] = (float)actions[env * 2 + 1];
        o_reward[env] = (float)reward[env];
        o_done[env] = (float)done[env];
        o_valid[env] = active ? (float)(active[env] != 0) : 1.0f;
    }
}

// MPPPO.py:301-312: G_t = r_t + gamma * G_{t+1}, walking the episode backwards.
// The reference does this arithmetic on f32 tensor elements (rewards come out of
// Buffer.sample() as float32; `discount_rate * tensor` and `+` are separate f32
// ops), so the scan is f32 with two roundings per step.  Rows that are not
// valid (env finished earlier) are skipped and get a 0 return.
__global__ void returns_kernel(int T, int N, float gamma, const float *reward, const float *valid, float *returns) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    float g = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)t * N + env;
        if (valid[o] != 0.0f) {
            g = __fadd_rn(reward[o], __fmul_rn(gamma, g));
            returns[o] = g;
        } else {
            returns[o] = 0.0f;
        }
    }
}

// returns_kernel followed by the reference's per-episode normalisation (MPPPO.py:258-261: min-max to [0, 1], then
// standardisation with the unbiased std), one thread per env = per episode: G - min, / (max - min + 1e-8); mean and
// variance over the valid rows; (G - mean) / (std + 1e-8); rows that are not valid get 0.  f32 like the tensor ops
// it replaces (~35 small launches per round); sums run over t in ascending order.
__global__ void returns_normalise_kernel(int T, int N, float gamma, int normalized, int standardized, const float *reward, const float *valid,
                                         float *returns, float *out) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    float g = 0.0f, gmin = 3.402823466e+38f, gmax = -3.402823466e+38f;
    int cnt = 0;
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)t * N + env;
        if (valid[o] != 0.0f) {
            g = __fadd_rn(reward[o], __fmul_rn(gamma, g));
            returns[o] = g;
            gmin = fminf(gmin, g); gmax = fmaxf(gmax, g);
            ++cnt;
        } else {
            returns[o] = 0.0f;
        }
    }
    const float span = __fadd_rn(__fsub_rn(gmax, gmin), 1e-8f);
    float sum = 0.0f;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * N + env;
        if (valid[o] != 0.0f) {
            float v = returns[o];
            if (normalized) v = __fdiv_rn(__fsub_rn(v, gmin), span);
            out[o] = v;
            sum = __fadd_rn(sum, v);
        } else {
            out[o] = 0.0f;
        }
    }
    if (!standardized) return;
    const float n = (float)(cnt > 1 ? cnt : 1), n1 = (float)(cnt - 1 > 1 ? cnt - 1 : 1);
    const float mean = __fdiv_rn(sum, n);
    float ss = 0.0f;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * N + env;
        if (valid[o] != 0.0f) { const float d = __fsub_rn(out[o], mean); ss = __fadd_rn(ss, __fmul_rn(d, d)); }
    }
    const float denom = __fadd_rn(sqrtf(__fdiv_rn(ss, n1)), 1e-8f);
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * N + env;
        if (valid[o] != 0.0f) out[o] = __fdiv_rn(__fsub_rn(out[o], mean), denom);
    }
}

// The same arithmetic, same order, for episodes of at most 64 steps (MPPPO's max_steps is 56): the episode's rewards and
// validity flags are loaded once with all loads in flight, the four passes run in registers, one store per row.  The
// kernel above walks memory four times with one thread per environment -- 0.12 ms of a 5.9 ms PPO round at 4096 envs.
constexpr int kEpisodeRegs = 64;
__global__ __launch_bounds__(64) void returns_normalise_regs_kernel(int T, int N, float gamma, int normalized, int standardized, const float *reward,
                                                                   const float *valid, float *returns, float *out) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    float r[kEpisodeRegs], v[kEpisodeRegs];
#pragma unroll
    for (int t = 0; t < kEpisodeRegs; ++t) {
        r[t] = 0.0f; v[t] = 0.0f;
        if (t < T) { r[t] = reward[(size_t)t * N + env]; v[t] = valid[(size_t)t * N + env]; }
    }
    float g = 0.0f, gmin = 3.402823466e+38f, gmax = -3.402823466e+38f;
    int cnt = 0;
#pragma unroll
    for (int t = kEpisodeRegs - 1; t >= 0; --t) {
        if (t < T) {
            if (v[t] != 0.0f) {
                g = __fadd_rn(r[t], __fmul_rn(gamma, g));
                r[t] = g;
                gmin = fminf(gmin, g); gmax = fmaxf(gmax, g);
                ++cnt;
            } else {
                r[t] = 0.0f;
            }
            returns[(size_t)t * N + env] = r[t];
        }
    }
    const float span = __fadd_rn(__fsub_rn(gmax, gmin), 1e-8f);
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < kEpisodeRegs; ++t) {
        if (t < T && v[t] != 0.0f) {
            if (normalized) r[t] = __fdiv_rn(__fsub_rn(r[t], gmin), span);
            sum = __fadd_rn(sum, r[t]);
        }
    }
    if (standardized) {
        const float n = (float)(cnt > 1 ? cnt : 1), n1 = (float)(cnt - 1 > 1 ? cnt - 1 : 1);
        const float mean = __fdiv_rn(sum, n);
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < kEpisodeRegs; ++t)
            if (t < T && v[t] != 0.0f) { const float d = __fsub_rn(r[t], mean); ss = __fadd_rn(ss, __fmul_rn(d, d)); }
        const float denom = __fadd_rn(sqrtf(__fdiv_rn(ss, n1)), 1e-8f);
#pragma unroll
        for (int t = 0; t < kEpisodeRegs; ++t)
            if (t < T && v[t] != 0.0f) r[t] = __fdiv_rn(__fsub_rn(r[t], mean), denom);
    }
#pragma unroll
    for (int t = 0; t < kEpisodeRegs; ++t)
        if (t < T) out[(size_t)t * N + env] = v[t] != 0.0f ? r[t] : 0.0f;
}

bool ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fjsp::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}


// pick_action_and_log_prob (agents/MPPPO/MPPPO.py:272-284) for one vector step, one thread per env: the sampler of
// fjsp_policy.h (shared with the fused policy rollout), and the action in the environment's encoding (pair
// (a / div, a % div), or (a, 0) for the flat-action environments).  Replaces the ~15 small launches torch needs for
// the same thing.
__global__ void policy_sample_kernel(const float *probs, int N, int A, int div, const float *epsilon, const uint64_t *seed,
                                     uint64_t counter, uint8_t *pair, float *action_out, float *logp_out) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    const fjsp::SampledAction sa = fjsp::sample_action(probs + (size_t)env * A, A, epsilon[0], seed[0], counter, env);
    logp_out[env] = sa.log_prob;
    action_out[env] = (float)sa.action;
    pair[env * 2] = (uint8_t)(div > 0 ? sa.action / div : sa.action);
    pair[env * 2 + 1] = (uint8_t)(div > 0 ? sa.action % div : 0);
}

}  // namespace

extern "C" {

int fjsp_rollout_create(int32_t T, int32_t N, int32_t state_size, int32_t device, fjsp_rollout **out) {
    if (T <= 0 || N <= 0 || state_size <= 0 || !out) { fjsp::set_error("fjsp_rollout_create: bad arguments"); return FJSP_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        fjsp::set_error("fjsp_rollout_create: no such HIP device"); return FJSP_E_HIP;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    auto *b = new fjsp_rollout();
    b->T = T; b->N = N; b->S = state_size; b->device = device;
    const size_t tn = (size_t)T * N;
    bool good = ok(hipMalloc((void **)&b->states, tn * state_size * 4), "hipMalloc states") &&
                ok(hipMalloc((void **)&b->next_states, tn * state_size * 4), "hipMalloc next_states") &&
                ok(hipMalloc((void **)&b->actions, tn * 2 * 4), "hipMalloc actions") &&
                ok(hipMalloc((void **)&b->rewards, tn * 4), "hipMalloc rewards") &&
                ok(hipMalloc((void **)&b->dones, tn * 4), "hipMalloc dones") &&
                ok(hipMalloc((void **)&b->valid, tn * 4), "hipMalloc valid") &&
                ok(hipMalloc((void **)&b->returns, tn * 4), "hipMalloc returns");
    if (good) good = ok(hipMemset(b->valid, 0, tn * 4), "hipMemset") && ok(hipMemset(b->returns, 0, tn * 4), "hipMemset");
    (void)hipSetDevice(prev);
    if (!good) { fjsp_rollout_destroy(b); return FJSP_E_HIP; }
    *out = b;
    return FJSP_OK;
}

void fjsp_rollout_destroy(fjsp_rollout *b) {
    if (!b) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(b->device);
    for (float *p : {b->states, b->actions, b->rewards, b->next_states, b->dones, b->valid, b->returns})
        if (p) (void)hipFree(p);
    (void)hipSetDevice(prev);
    delete b;
}

int fjsp_rollout_append(fjsp_rollout *b, const double *d_state, const uint8_t *d_actions, const double *d_reward,
                        const double *d_next_state, const uint8_t *d_done, const uint8_t *d_active, void *stream) {
    if (!b || !d_state || !d_actions || !d_reward || !d_next_state || !d_done) { fjsp::set_error("fjsp_rollout_append: null argument"); return FJSP_E_ARG; }
    if (b->len >= b->T) { fjsp::set_error("fjsp_rollout_append: buffer full"); return FJSP_E_STATE; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(b->device);
    const size_t t = (size_t)b->len, N = (size_t)b->N, S = (size_t)b->S;
    const size_t n = N * S;
    hipLaunchKernelGGL(append_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->N, b->S,
                       d_state, d_actions, d_reward, d_next_state, d_done, d_active, b->states + t * n,
                       b->actions + t * N * 2, b->rewards + t * N, b->next_states + t * n, b->dones + t * N,
                       b->valid + t * N);
    const bool good = ok(hipGetLastError(), "append_kernel");
    (void)hipSetDevice(prev);
    if (!good) return FJSP_E_HIP;
    b->len++;
    return FJSP_OK;
}

int fjsp_rollout_returns(fjsp_rollout *b, double gamma, void *stream) {
    if (!b) { fjsp::set_error("fjsp_rollout_returns: null buffer"); return FJSP_E_ARG; }
    if (b->len == 0) return FJSP_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(b->device);
    hipLaunchKernelGGL(returns_kernel, dim3((unsigned)((b->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->len,
                       b->N, (float)gamma, b->rewards, b->valid, b->returns);
    const bool good = ok(hipGetLastError(), "returns_kernel");
    (void)hipSetDevice(prev);
    return good ? FJSP_OK : FJSP_E_HIP;
}

int fjsp_rollout_returns_normalised(fjsp_rollout *b, double gamma, int32_t normalized, int32_t standardized, float *d_out, void *stream) {
    if (!b || !d_out) { fjsp::set_error("fjsp_rollout_returns_normalised: null argument"); return FJSP_E_ARG; }
    if (b->len == 0) return FJSP_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(b->device);
    if (b->len <= kEpisodeRegs)
        hipLaunchKernelGGL(returns_normalise_regs_kernel, dim3((unsigned)((b->N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, b->len, b->N,
                           (float)gamma, (int)normalized, (int)standardized, b->rewards, b->valid, b->returns, d_out);
    else
        hipLaunchKernelGGL(returns_normalise_kernel, dim3((unsigned)((b->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->len, b->N, (float)gamma,
                           (int)normalized, (int)standardized, b->rewards, b->valid, b->returns, d_out);
    const bool good = ok(hipGetLastError(), "returns_normalise_kernel");
    (void)hipSetDevice(prev);
    return good ? FJSP_OK : FJSP_E_HIP;
}

int fjsp_rollout_clear(fjsp_rollout *b) {
    if (!b) { fjsp::set_error("fjsp_rollout_clear: null buffer"); return FJSP_E_ARG; }
    b->len = 0;
    return FJSP_OK;
}
int fjsp_rollout_len(const fjsp_rollout *b) { return b ? b->len : 0; }

void *fjsp_rollout_ptr(fjsp_rollout *b, int32_t which) {
    if (!b) return nullptr;
    switch (which) {
    case 0: return b->states;
    case 1: return b->actions;
    case 2: return b->rewards;
    case 3: return b->next_states;
    case 4: return b->dones;
    case 5: return b->valid;
    case 6: return b->returns;
    default: return nullptr;
    }
}

}  // extern "C"

extern "C" int fjsp_policy_sample(const float *d_probs, int32_t n, int32_t n_actions, int32_t pair_div, const float *d_epsilon,
                                  const uint64_t *d_seed, uint64_t counter, uint8_t *d_pair, float *d_action, float *d_log_prob,
                                  void *stream) {
    if (!d_probs || !d_epsilon || !d_seed || !d_pair || !d_action || !d_log_prob || n <= 0 || n_actions <= 0 || n_actions > 256 ||
        pair_div < 0) {
        fjsp::set_error("fjsp_policy_sample: bad arguments"); return FJSP_E_ARG;
    }
    hipLaunchKernelGGL(policy_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_probs, (int)n,
                       (int)n_actions, (int)pair_div, d_epsilon, d_seed, counter, d_pair, d_action, d_log_prob);
    if (hipGetLastError() != hipSuccess) { fjsp::set_error("policy_sample_kernel launch failed"); return FJSP_E_HIP; }
    return FJSP_OK;
}
