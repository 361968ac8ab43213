// Device code shared by the policy-side kernels: the categorical sampler of fjsp_policy_sample
// (fjsp_rollout_buffer.hip) and the 2-hidden-layer actor MLP that the fused policy rollout evaluates inside the
// environment kernel (fjsp_kernels.hip).  Both exist ONCE, here, so that the per-step path (actor kernel ->
// sampler kernel -> step kernel) and the fused path (one launch per rollout) perform the same arithmetic in the
// same order and produce the same actions bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fjsp {

__device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// pick_action_and_log_prob (agents/MPPPO/MPPPO.py:272-284) for ONE environment, by one thread:
// Categorical(probs).sample() by inverse CDF over the normalised probabilities (sums strictly in action order), the
// epsilon-random override (:279-280), log_prob of the taken action with torch's clamp of the normalised probability to
// [eps, 1 - eps].  `p` may point to global memory or LDS.
struct SampledAction { int action; float log_prob; };
__device__ inline SampledAction sample_action(const float *p, int A, float epsilon, uint64_t seed, uint64_t counter, int env) {
    float total = 0.0f;
    for (int a = 0; a < A; ++a) total += p[a];
    const uint64_t r = mix64(seed ^ mix64(counter * 0x100000001B3ULL + (uint64_t)env));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);                 // [0, 1)
    const float v = (float)((r >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
    int action = A - 1;
    float acc = 0.0f;
    const float target = u * total;
    for (int a = 0; a < A; ++a) {
        acc += p[a];
        if (acc > target) { action = a; break; }
    }
    if (v <= epsilon) action = (int)(mix64(r) % (uint64_t)A);                  // random.randint(0, A - 1)
    float pn = p[action] / total;
    pn = fminf(fmaxf(pn, 1.1920929e-07f), 1.0f - 1.1920929e-07f);
    SampledAction out;
    out.action = action;
    out.log_prob = logf(pn);
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Actor MLP  S -> 128 -> 128 -> A  (ActorNet, agents/MPPPO/MPPPO.py:31-48: Linear + ReLU, Linear + ReLU, Linear,
// softmax) for ONE state per wavefront, weights in LDS (shared by the 16 waves of the workgroup).
//
// LDS weight image (floats): W1T[S][64][2], W2T[128][64][2] -- entry (i, lane) holds the weights of input i for
// hidden units `lane` and `lane + 64`, one 8-byte read per lane and input; W3T[128][32] (actions padded to 32 with
// zero columns); b1[128], b2[128], b3[32].  Lane l owns hidden units l and l + 64; the output layer splits its 128
// inputs between lanes a (inputs 0..63) and a + 32 (inputs 64..127) of action a.  Every sum is an fmaf chain in a
// fixed order, so the result is a pure function of (weights, state) wherever it is evaluated.
constexpr int kActorH = 128;
constexpr int kActorAP = 32;
struct ActorParams {            // device pointers, torch layouts: W[out][in] row-major
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    int S, H, A;
};
__host__ __device__ inline size_t actor_lds_floats(int S) {
    return (size_t)S * 128 + 128 * 128 + 128 * kActorAP + 128 + 128 + kActorAP;
}
struct ActorLds {
    const float *w1t, *w2t, *w3t, *b1, *b2, *b3;
};
__device__ inline ActorLds actor_lds_carve(float *base, int S) {
    ActorLds a;
    a.w1t = base; base += (size_t)S * 128;
    a.w2t = base; base += 128 * 128;
    a.w3t = base; base += 128 * kActorAP;
    a.b1 = base; base += 128;
    a.b2 = base; base += 128;
    a.b3 = base;
    return a;
}
// every thread of the workgroup takes part; the caller synchronises the workgroup afterwards
__device__ inline void actor_lds_fill(float *base, const ActorParams &p, int tid, int nthreads) {
    const int S = p.S, A = p.A;
    float *w1t = base, *w2t = w1t + (size_t)S * 128, *w3t = w2t + 128 * 128, *b1 = w3t + 128 * kActorAP, *b2 = b1 + 128,
          *b3 = b2 + 128;
    for (int e = tid; e < S * 128; e += nthreads) {            // e = (i * 64 + lane) * 2 + half  ->  W1[lane + 64 half][i]
        const int half = e & 1, lane = (e >> 1) & 63, i = e >> 7;
        w1t[e] = p.w1[(size_t)(lane + 64 * half) * S + i];
    }
    for (int e = tid; e < 128 * 128; e += nthreads) {
        const int half = e & 1, lane = (e >> 1) & 63, i = e >> 7;
        w2t[e] = p.w2[(size_t)(lane + 64 * half) * 128 + i];
    }
    for (int e = tid; e < 128 * kActorAP; e += nthreads) {     // e = i * 32 + a  ->  W3[a][i]
        const int a = e & 31, i = e >> 5;
        w3t[e] = a < A ? p.w3[(size_t)a * 128 + i] : 0.0f;
    }
    for (int e = tid; e < 128; e += nthreads) { b1[e] = p.b1[e]; b2[e] = p.b2[e]; }
    for (int e = tid; e < kActorAP; e += nthreads) b3[e] = e < A ? p.b3[e] : 0.0f;
}

// Where the fused policy rollout writes one vector step: the rows of the on-policy buffer (fjsp_rollout:
// Buffer.py:19-28 `.float()` conversions) plus what FusedSampler keeps of the sampled actions.
struct PolicyRolloutIO {
    const double *state_in;      // [N][S] the states the rollout starts from (reset output)
    const float *epsilon;        // [1]   exploration rate of the round
    const uint64_t *seed;        // [1]   sampling stream of the round
    int pair_div;                // > 0: action -> (a / div, a % div); 0: flat action
    float *o_state, *o_actions, *o_reward, *o_next, *o_done, *o_valid;    // [T][N][...]
    float *o_flat, *o_logp;      // [T][N] flat action index, log-probability of the taken action
    double *state_last;          // [N][S] state after the last step of every env (the batch's state buffer)
};

// LDS-only hand-off between the lanes of one wave (see wave_sync() in fjsp_kernels.hip)
__device__ __forceinline__ void actor_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// x: S floats in LDS (wave-private); h: 128 floats of wave-private LDS scratch; probs: kActorAP floats of
// wave-private LDS, receives softmax(logits)[0..A).  All 64 lanes of the wave call this.
// `wbase`: the LDS weight image (actor_lds_fill); carved here so that callers keep one pointer live, not six
__device__ inline void actor_probs(const float *wbase, const float *x, float *h, float *probs, int S, int A) {
    const ActorLds w = actor_lds_carve(const_cast<float *>(wbase), S);
    const int lane = (int)__lane_id();
    // ---- layer 1
    float a0 = w.b1[lane], a1 = w.b1[lane + 64];
    for (int i = 0; i < S; ++i) {
        const float2 wv = reinterpret_cast<const float2 *>(w.w1t)[i * 64 + lane];
        const float xi = x[i];
        a0 = __fmaf_rn(xi, wv.x, a0); a1 = __fmaf_rn(xi, wv.y, a1);
    }
    h[lane] = fmaxf(a0, 0.0f); h[lane + 64] = fmaxf(a1, 0.0f);
    actor_wave_sync();
    // ---- layer 2 (inputs fetched four at a time: one broadcast read)
    a0 = w.b2[lane]; a1 = w.b2[lane + 64];
    for (int i = 0; i < 128; i += 4) {
        const float4 hv = *reinterpret_cast<const float4 *>(h + i);
        const float2 w0 = reinterpret_cast<const float2 *>(w.w2t)[(i + 0) * 64 + lane];
        const float2 w1 = reinterpret_cast<const float2 *>(w.w2t)[(i + 1) * 64 + lane];
        const float2 w2 = reinterpret_cast<const float2 *>(w.w2t)[(i + 2) * 64 + lane];
        const float2 w3 = reinterpret_cast<const float2 *>(w.w2t)[(i + 3) * 64 + lane];
        a0 = __fmaf_rn(hv.x, w0.x, a0); a1 = __fmaf_rn(hv.x, w0.y, a1);
        a0 = __fmaf_rn(hv.y, w1.x, a0); a1 = __fmaf_rn(hv.y, w1.y, a1);
        a0 = __fmaf_rn(hv.z, w2.x, a0); a1 = __fmaf_rn(hv.z, w2.y, a1);
        a0 = __fmaf_rn(hv.w, w3.x, a0); a1 = __fmaf_rn(hv.w, w3.y, a1);
    }
    actor_wave_sync();                                         // everybody has read h1 before it is overwritten
    h[lane] = fmaxf(a0, 0.0f); h[lane + 64] = fmaxf(a1, 0.0f);
    actor_wave_sync();
    // ---- output layer: lane a sums inputs 0..63, lane a + 32 inputs 64..127 (the bias rides on the first half)
    const int act = lane & 31, half = lane >> 5;
    float z = half == 0 ? w.b3[act] : 0.0f;
    const float *hh = h + 64 * half;
    const float *wc = w.w3t + (size_t)(64 * half) * kActorAP + act;
    for (int i = 0; i < 64; i += 4) {
        const float4 hv = *reinterpret_cast<const float4 *>(hh + i);
        z = __fmaf_rn(hv.x, wc[(i + 0) * kActorAP], z);
        z = __fmaf_rn(hv.y, wc[(i + 1) * kActorAP], z);
        z = __fmaf_rn(hv.z, wc[(i + 2) * kActorAP], z);
        z = __fmaf_rn(hv.w, wc[(i + 3) * kActorAP], z);
    }
    // the second half joins the first: lane a reads lane a + 32
    const float other = __shfl(z, (lane + 32) & 63, 64);
    float logit = z + other;                                   // (lanes >= 32 compute the same value, unused)
    // ---- softmax over the A actions (F.softmax: exp(z - max) / sum), sum strictly in action order
    const bool is_act = lane < A;
    float m = is_act ? logit : -INFINITY;
    for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));      // lanes 0..31 hold every action
    m = __shfl(m, 0, 64);
    const float e = is_act ? expf(logit - m) : 0.0f;
    if (lane < kActorAP) probs[lane] = e;
    actor_wave_sync();
    float sum = 0.0f;
    for (int a = 0; a < A; ++a) sum += probs[a];               // (uniform: every lane walks the same broadcast reads)
    actor_wave_sync();
    if (lane < kActorAP) probs[lane] = is_act ? e / sum : 0.0f;
    actor_wave_sync();
}

}  // namespace fjsp
