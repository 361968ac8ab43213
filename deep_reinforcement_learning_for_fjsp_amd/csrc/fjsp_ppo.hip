// Fused pieces of the clipped-PPO learning iteration (agents/MPPPO/MPPPO.py:314-370) behind the C ABI.
//
// The dense layers of the 2 x 128 actor / critic stay on the library GEMMs (MFMA); what this file replaces is the
// swarm of small element-wise / reduction launches between them -- ~50 per network and iteration in eager PyTorch,
// a third of a learning round's device time (profiles/r02_ppo_round_*): the loss with its gradient in ONE pass over
// the logits, the ReLU backward together with the bias gradient in ONE pass over a hidden activation, and the
// clip-by-global-norm + Adam step over ONE flat parameter buffer.
//
//   fjsp_ppo_actor_loss      log_softmax + gather, ratio = exp(new) / (exp(old) + 1e-8), clipped surrogate,
//                            -mean over the samples (:325-352), and d loss / d logits
//   fjsp_ppo_critic_loss     F.mse_loss(critic(s), G) (:317-318) and d loss / d value
//   fjsp_relu_bwd_bias       dz = dh * (h > 0) in place, per-block column sums of dz (bias gradient partials)
//   fjsp_col_sum_finish      adds the partials up (deterministic two-stage reduction, no atomics)
//   fjsp_adam_clip_step      torch.nn.utils.clip_grad_norm_ (:362,369) + Adam (lr, eps 1e-4, :145-147) on flat buffers
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/fjsp_amd.h"
#include "fjsp_host.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;

__device__ inline float block_sum(float v, float *sh) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;       // valid in thread 0
}

// 32 lanes per sample (n_actions <= 32: lane a holds logit a; two samples per wavefront), reductions by xor-shuffles
// inside the 32-lane half.  The sums run in shuffle-tree order (a fixed order: the result is deterministic).
__global__ __launch_bounds__(256) void actor_loss_kernel(const float *logits, const float *actions, const float *old_logp, const float *adv,
                                                         int n, int A, float clip_eps, const float *count, float *dlogits, float *loss_partial) {
    __shared__ float sh[kBlock / 64];
    const int a = threadIdx.x & 31;
    const int i = blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);
    float term = 0.0f;
    const bool row = i < n, on = row && a < A;
    const float z = on ? logits[(size_t)i * A + a] : -INFINITY;
    float m = z;
    for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float e = on ? expf(z - m) : 0.0f;
    float s = e;
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (row) {
        const float lse = m + logf(s);
        const int act = (int)actions[i];
        const float z_act = __shfl(z, (threadIdx.x & 32) + act, 64);
        const float new_lp = z_act - lse;                                      // log_softmax(...).gather(action), :327-328
        const float ratio = expf(new_lp) / (expf(old_logp[i]) + 1e-8f);         // :330-333
        const float lo = 1.0f - clip_eps, hi = 1.0f + clip_eps;
        const float clipped = fminf(fmaxf(ratio, lo), hi);
        const float ad = adv[i];
        const float s1 = ad * ratio, s2 = ad * clipped;                         // :344-350
        // d term / d ratio with autograd's conventions: minimum splits a tie in half, clamp passes the gradient on [lo, hi]
        const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        float g;
        if (s1 < s2) g = ad;
        else if (s2 < s1) g = ad * inside;
        else g = 0.5f * ad + 0.5f * ad * inside;
        // loss = -sum(term) / count; d ratio / d new_lp = ratio; d new_lp / d z_j = [j == act] - softmax_j
        const float c = -g * ratio / count[0];
        if (on) dlogits[(size_t)i * A + a] = c * ((a == act ? 1.0f : 0.0f) - e / s);
        if (a == 0) term = fminf(s1, s2);
    }
    const float t = block_sum(term, sh);
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = t;
}

__global__ void critic_loss_kernel(const float *value, const float *returns, int n, const float *count, float *dvalue, float *loss_partial) {
    __shared__ float sh[kBlock / 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.0f;
    if (i < n) {
        const float d = value[i] - returns[i];
        term = d * d;
        dvalue[i] = 2.0f * d / count[0];
    }
    const float t = block_sum(term, sh);
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = t;
}

// loss = sign * sum(partials) / count  (one block)
__global__ void loss_finish_kernel(const float *partial, int nparts, const float *count, float sign, float *loss) {
    __shared__ float sh[kBlock / 64];
    float v = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) v += partial[i];
    const float t = block_sum(v, sh);
    if (threadIdx.x == 0) loss[0] = sign * t / count[0];
}

// dz = dh * (h > 0) in place (h == nullptr: no mask) and column sums, two stages.  A block owns a contiguous band
// of rows; its 256 threads form RL row lanes x CT column threads (CT = columns / VEC, VEC = 4 floats per thread when
// the width allows: 16-byte accesses, a row of 128 floats is one 512-byte burst); the row lanes are folded through
// LDS into partial[block][width].
template <int VEC>
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(float *dh, const float *h, int n, int width, int rows_per_block, float *partial) {
    __shared__ float sh[256 * VEC];
    const int ct = width / VEC;                       // column threads per row lane (<= 256)
    const int rl = 256 / ct;                          // row lanes
    const int cid = threadIdx.x % ct, rid = threadIdx.x / ct;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
    if (rid < rl) {
        for (int r = r0 + rid; r < r1; r += rl) {
            const size_t o = (size_t)r * width + (size_t)cid * VEC;
            if constexpr (VEC == 4) {
                float4 g = *reinterpret_cast<const float4 *>(dh + o);
                if (h) {
                    const float4 a = *reinterpret_cast<const float4 *>(h + o);
                    g.x = a.x > 0.0f ? g.x : 0.0f; g.y = a.y > 0.0f ? g.y : 0.0f; g.z = a.z > 0.0f ? g.z : 0.0f; g.w = a.w > 0.0f ? g.w : 0.0f;
                    *reinterpret_cast<float4 *>(dh + o) = g;
                }
                acc[0] += g.x; acc[1] += g.y; acc[2] += g.z; acc[3] += g.w;
            } else {
                float g = dh[o];
                if (h) { g = h[o] > 0.0f ? g : 0.0f; dh[o] = g; }
                acc[0] += g;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) sh[threadIdx.x * VEC + v] = (rid < rl) ? acc[v] : 0.0f;
    __syncthreads();
    if (threadIdx.x < ct) {
        for (int v = 0; v < VEC; ++v) {
            float s = 0.0f;
            for (int q = 0; q < rl; ++q) s += sh[(q * ct + threadIdx.x) * VEC + v];
            partial[(size_t)blockIdx.x * width + (size_t)threadIdx.x * VEC + v] = s;
        }
    }
}
// out[col] = sum over the partial rows: one block per column, the block folds its column through LDS
__global__ __launch_bounds__(256) void col_sum_finish_kernel(const float *partial, int nblocks, int width, float *out) {
    __shared__ float sh[kBlock / 64];
    const int col = blockIdx.x;
    float s = 0.0f;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partial[(size_t)b * width + col];
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) out[col] = t;
}

// sum of squares of a flat buffer -> partial[blocks]
__global__ void sumsq_kernel(const float *g, int n, float *partial) {
    __shared__ float sh[kBlock / 64];
    float v = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += g[i] * g[i];
    const float t = block_sum(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// clip coefficient (clip_grad_norm_: max_norm / (total_norm + 1e-6), clamped to 1) and the Adam update; the step
// count lives on the device (captured graphs replay the launch)
__global__ void adam_clip_kernel(float *p, const float *g, float *m, float *v, int n, const float *sumsq_partial, int nparts, float max_norm,
                                 float lr, float beta1, float beta2, float eps, float *step) {
    __shared__ float coef_s, bc1_s, bc2_s;
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (int i = 0; i < nparts; ++i) ss += sumsq_partial[i];
        const float total = sqrtf(ss);
        coef_s = max_norm > 0.0f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;
        const float t = step[0] + 1.0f;                       // (every block reads the old count; block 0 writes the new one below)
        bc1_s = 1.0f - powf(beta1, t);
        bc2_s = 1.0f - powf(beta2, t);
    }
    __syncthreads();
    const float coef = coef_s, bc1 = bc1_s, bc2 = bc2_s;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;             // torch.optim.Adam (no amsgrad, no weight decay)
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}
__global__ void step_inc_kernel(float *step) { step[0] += 1.0f; }

bool launched(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return true;
    fjsp::set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

}  // namespace

extern "C" {

int fjsp_ppo_partials(int32_t n) { return (n + kBlock / 32 - 1) / (kBlock / 32); }

int fjsp_ppo_actor_loss(const float *d_logits, const float *d_actions, const float *d_old_log_prob, const float *d_advantages, int32_t n,
                        int32_t n_actions, float clip_epsilon, const float *d_count, float *d_dlogits, float *d_partial, float *d_loss,
                        void *stream) {
    if (!d_logits || !d_actions || !d_old_log_prob || !d_advantages || !d_count || !d_dlogits || !d_partial || !d_loss || n <= 0 ||
        n_actions <= 0 || n_actions > 32) { fjsp::set_error("fjsp_ppo_actor_loss: bad arguments (n_actions <= 32)"); return FJSP_E_ARG; }
    const int blocks = (n + kBlock / 32 - 1) / (kBlock / 32);
    hipLaunchKernelGGL(actor_loss_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, d_logits, d_actions, d_old_log_prob,
                       d_advantages, n, n_actions, clip_epsilon, d_count, d_dlogits, d_partial);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, d_partial, blocks, d_count, -1.0f, d_loss);
    return launched("fjsp_ppo_actor_loss") ? FJSP_OK : FJSP_E_HIP;
}

int fjsp_ppo_critic_loss(const float *d_value, const float *d_returns, int32_t n, const float *d_count, float *d_dvalue, float *d_partial,
                         float *d_loss, void *stream) {
    if (!d_value || !d_returns || !d_count || !d_dvalue || !d_partial || !d_loss || n <= 0) { fjsp::set_error("fjsp_ppo_critic_loss: bad arguments"); return FJSP_E_ARG; }
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(critic_loss_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, d_value, d_returns, n, d_count, d_dvalue, d_partial);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, d_partial, blocks, d_count, 1.0f, d_loss);
    return launched("fjsp_ppo_critic_loss") ? FJSP_OK : FJSP_E_HIP;
}

int fjsp_relu_bwd_bias(float *d_dh, const float *d_h, int32_t n, int32_t width, float *d_partial, int32_t n_partial_rows, float *d_bias_grad,
                       void *stream) {
    if (!d_dh || !d_partial || !d_bias_grad || n <= 0 || width <= 0 || width > 256 || n_partial_rows <= 0) {
        fjsp::set_error("fjsp_relu_bwd_bias: bad arguments (width <= 256)"); return FJSP_E_ARG;
    }
    const int rows = (n + n_partial_rows - 1) / n_partial_rows;
    const int blocks = (n + rows - 1) / rows;
    const bool vec = width % 4 == 0 && (reinterpret_cast<uintptr_t>(d_dh) & 15) == 0 && (!d_h || (reinterpret_cast<uintptr_t>(d_h) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(relu_bwd_bias_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_dh, d_h, n, width, rows, d_partial);
    else
        hipLaunchKernelGGL(relu_bwd_bias_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_dh, d_h, n, width, rows, d_partial);
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3(width), dim3(256), 0, (hipStream_t)stream, d_partial, blocks, width, d_bias_grad);
    return launched("fjsp_relu_bwd_bias") ? FJSP_OK : FJSP_E_HIP;
}

int fjsp_adam_clip_step(float *d_params, const float *d_grads, float *d_exp_avg, float *d_exp_avg_sq, int32_t n, float max_norm, float lr,
                        float beta1, float beta2, float eps, float *d_step, float *d_scratch64, void *stream) {
    if (!d_params || !d_grads || !d_exp_avg || !d_exp_avg_sq || !d_step || !d_scratch64 || n <= 0) { fjsp::set_error("fjsp_adam_clip_step: bad arguments"); return FJSP_E_ARG; }
    const int parts = 64;
    hipLaunchKernelGGL(sumsq_kernel, dim3(parts), dim3(kBlock), 0, (hipStream_t)stream, d_grads, n, d_scratch64);
    const int blocks = std::min(256, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, d_params, d_grads, d_exp_avg, d_exp_avg_sq, n,
                       d_scratch64, parts, max_norm, lr, beta1, beta2, eps, d_step);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_step);
    return launched("fjsp_adam_clip_step") ? FJSP_OK : FJSP_E_HIP;
}

}  // extern "C"
