// One launch = forward + loss + backward of a Linear(S,128)-ReLU-Linear(128,128)-ReLU-Linear(128,A) network over
// the whole sample set of a clipped-PPO learning iteration (agents/MPPPO/MPPPO.py:314-370; actor: :325-352, critic:
// :317-318), on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, one rounding per product).
//
// The library-GEMM trainer (agents/fused_mlp.py `backward()`) writes every hidden activation and every activation
// gradient to HBM and reads it back two or three times: ~1.3 GB per pass for 164 k samples, 20 passes per round.  Here a
// workgroup (4 wavefronts, one per SIMD, one workgroup per CU) keeps a tile of 32 samples in LDS from the input row to
// the weight-gradient products; HBM sees the input rows once (S floats per sample), the per-sample loss inputs, and at
// the end one partial gradient per workgroup.
//
//   LDS (124 KB):  W2 [128][132] | X tile [32][36] (column S = 1: carries b1 and its gradient through the products)
//                  H1 [32][132] -> dH1 in place | H2 [32][132] -> dH2 in place | 4 split-K partials of the logits |
//                  d logits [32][36]
//   registers:     W1 (+b1) and W3 fragments of the wave's 32 output columns / K quarter (48), the weight-gradient
//                  accumulators dW2 (64), dW1 (16), dW3 (16), bias-gradient partial sums
//
// Products per tile (M = 32 samples on the rows of a 32x32 MFMA tile; wave w owns output columns 32w..32w+31, or the
// K quarter 32w.. for the logits), 272 MFMAs per wave:
//   G1 H1 = relu([X 1] [W1 b1]^T)           16      G5 dH2 = (dOUT W3) * (H2 > 0)            16
//   G2 H2 = relu(H1 W2^T + b2)              64      G6 dH1 = (dH2 W2) * (H1 > 0)             64
//   G3 OUT = H2 W3^T + b3 (split-K)         16      G7 dW2 += dH2^T H1                       64
//   G4 dW3 += dOUT^T H2                     16      G8 [dW1 db1] += dH1^T [X 1]              16
// Operand fragments: a chunk of 8 k-values feeds 4 MFMAs; lane half h supplies k = 8 chunk + 4 h + j to MFMA j, so an
// operand that is k-contiguous in LDS is ONE ds_read_b128 per chunk (rows padded to 132 / 36 floats: conflict-free),
// a k-strided operand (the transposed uses of W2, H1, H2, X, dOUT) is 4 ds_read_b32 of 32 consecutive floats.
//
// Sums over the samples run in a fixed order (tile order inside a workgroup, then workgroup order in the finish
// kernel): results are deterministic, and differ from the library path by f32 reassociation only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/fjsp_amd.h"
#include "fjsp_host.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 128;        // hidden width (the reference's 2 x 128 networks; other widths use the library path)
constexpr int HS = 132;       // LDS row stride of the 128-wide images
constexpr int XS = 36;        // LDS row stride of the 32-wide images
constexpr int TS = 32;        // samples per tile
constexpr int kLdsFloats = H * HS + TS * XS + 2 * TS * HS + 4 * TS * 32 + TS * XS + 256 + 8;

struct PassArgs {
    const float *params;      // flat: W1[H][S] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]
    const float *x;           // [n][S]
    const float *aux0;        // actor: actions (f32) | critic: returns
    const float *aux1;        // actor: old log-probabilities
    const float *aux2;        // actor: advantages
    const float *count;       // [1] global sample count (the mean's denominator)
    float *partial;           // [gridDim.x][numel] gradient partials, parameter order
    float *loss_partial;      // [gridDim.x]
    int n, S, A, numel;
    float clip_eps;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

template <int MODE>   // 0: actor (clipped surrogate on softmax logits), 1: critic (mean squared error on the value)
__global__ __launch_bounds__(256) void mlp_train_pass_kernel(PassArgs p) {
    extern __shared__ float lds[];
    float *W2s = lds;
    float *Xs = W2s + H * HS;
    float *H1s = Xs + TS * XS;
    float *H2s = H1s + TS * HS;
    float *OUTp = H2s + TS * HS;
    float *dOs = OUTp + 4 * TS * 32;
    float *red = dOs + TS * XS;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, c = l & 31, hf = l >> 5;
    const int S = p.S, A = p.A, n = p.n;
    const float *W1 = p.params, *b1 = W1 + H * S, *W2 = b1 + H, *b2 = W2 + H * H, *W3 = b2 + H, *b3 = W3 + A * H;
    const int col = 32 * w + c;               // the output column / weight row this lane serves

    for (int i = tid; i < H * H / 4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(W2)[i];
        *reinterpret_cast<f32x4 *>(&W2s[(i >> 5) * HS + (i & 31) * 4]) = v;
    }
    // X image: column S = 1 (bias), columns above it 0; rewritten columns 0..S-1 per tile
    for (int i = tid; i < TS * 32; i += 256) Xs[(i >> 5) * XS + (i & 31)] = ((i & 31) == S) ? 1.0f : 0.0f;

    float w1f[16], w3f[16], w3t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = (q >> 2) * 8 + 4 * hf + (q & 3);
        w1f[q] = k < S ? W1[col * S + k] : (k == S ? b1[col] : 0.0f);           // B[k][o] of G1
        w3f[q] = c < A ? W3[c * H + 32 * w + k] : 0.0f;                        // B[k = h][a] of G3 (K quarter w)
        w3t[q] = k < A ? W3[k * H + col] : 0.0f;                               // B[k = a][h] of G5
    }
    const float b2c = b2[col];
    const float b3c = (w == 0 && c < A) ? b3[c] : 0.0f;
    const float inv_count = 1.0f / p.count[0];

    f32x16 dW2a[4], dW1a, dW3a;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW1a[r] = 0.0f; dW3a[r] = 0.0f; dW2a[0][r] = 0.0f; dW2a[1][r] = 0.0f; dW2a[2][r] = 0.0f; dW2a[3][r] = 0.0f; }
    float db2a = 0.0f, db3a = 0.0f, loss_a = 0.0f;

    // this thread's share of a tile's 32*S contiguous input floats
    int xdst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        xdst[q] = e < TS * S ? (e / S) * XS + (e % S) : -1;
    }
    const int ntiles = (n + TS - 1) / TS;
    float xr[4];
    auto load_x = [&](int tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t idx = (size_t)tile * TS * S + tid + 256 * q;
            xr[q] = (xdst[q] >= 0 && tile < ntiles && idx < (size_t)n * S) ? p.x[idx] : 0.0f;
        }
    };
    load_x(blockIdx.x);
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (xdst[q] >= 0) Xs[xdst[q]] = xr[q];
        // per-sample loss inputs of the 4 samples this lane's half-wave serves in the loss stage
        float m0[4], m1[4], m2[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int gs = tile * TS + it * 8 + 2 * w + hf;
            const bool v = gs < n;
            m0[it] = v ? p.aux0[gs] : 0.0f;
            if (MODE == 0) { m1[it] = v ? p.aux1[gs] : 0.0f; m2[it] = v ? p.aux2[gs] : 0.0f; }
        }
        load_x(tile + gridDim.x);
        __syncthreads();

        // ---- G1: H1 = relu([X 1] [W1 b1]^T) ------------------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const f32x4 a = ld4(&Xs[c * XS + kc * 8 + 4 * hf]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma(a[j], w1f[kc * 4 + j], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) H1s[acc_row(r, hf) * HS + col] = fmaxf(acc[r], 0.0f);
        __syncthreads();

        // ---- G2: H2 = relu(H1 W2^T + b2) ---------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b2c;
        {
            f32x4 a = ld4(&H1s[c * HS + 4 * hf]), b = ld4(&W2s[col * HS + 4 * hf]);
#pragma unroll
            for (int kc = 0; kc < 16; ++kc) {
                f32x4 an = a, bn = b;
                if (kc + 1 < 16) { an = ld4(&H1s[c * HS + (kc + 1) * 8 + 4 * hf]); bn = ld4(&W2s[col * HS + (kc + 1) * 8 + 4 * hf]); }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma(a[j], b[j], acc);
                a = an; b = bn;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) H2s[acc_row(r, hf) * HS + col] = fmaxf(acc[r], 0.0f);
        // (no barrier: G3 reads only the columns this wave has just written)

        // ---- G3: logits, K quarter w -> OUTp[w] ---------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b3c;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const f32x4 a = ld4(&H2s[c * HS + 32 * w + kc * 8 + 4 * hf]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma(a[j], w3f[kc * 4 + j], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) OUTp[w * (TS * 32) + acc_row(r, hf) * 32 + c] = acc[r];
        __syncthreads();

        // ---- loss and d loss / d logits: 32 lanes per sample (lane c = logit c), 8 samples per sweep ----------
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int s = it * 8 + 2 * w + hf;
            const bool valid = tile * TS + s < n;
            const bool on = valid && c < A;
            const float z = ((OUTp[s * 32 + c] + OUTp[TS * 32 + s * 32 + c]) + OUTp[2 * TS * 32 + s * 32 + c]) + OUTp[3 * TS * 32 + s * 32 + c];
            float d = 0.0f;
            if (MODE == 0) {
                const float zz = on ? z : -INFINITY;
                float m = zz;
                for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                const float e = on ? expf(zz - m) : 0.0f;
                float sum = e;
                for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
                const int act = (int)m0[it];
                const float z_act = __shfl(zz, (l & 32) + act, 64);
                if (valid) {
                    const float lse = m + logf(sum);
                    const float new_lp = z_act - lse;                                   // log_softmax(...).gather(action), :327-328
                    const float ratio = expf(new_lp) / (expf(m1[it]) + 1e-8f);            // :330-333
                    const float lo = 1.0f - p.clip_eps, hi = 1.0f + p.clip_eps;
                    const float clipped = fminf(fmaxf(ratio, lo), hi);
                    const float ad = m2[it];
                    const float s1 = ad * ratio, s2 = ad * clipped;                      // :344-350
                    const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;     // autograd: clamp passes the gradient on [lo, hi],
                    float g;                                                             // minimum splits a tie in half
                    if (s1 < s2) g = ad;
                    else if (s2 < s1) g = ad * inside;
                    else g = 0.5f * ad + 0.5f * ad * inside;
                    const float cf = -g * ratio * inv_count;
                    if (on) d = cf * ((c == act ? 1.0f : 0.0f) - e / sum);
                    if (c == 0) loss_a += fminf(s1, s2);
                }
            } else {
                if (on) {                                                               // c == 0: the value
                    const float dv = z - m0[it];
                    d = 2.0f * dv * inv_count;
                    loss_a += dv * dv;
                }
            }
            dOs[s * XS + c] = d;
            db3a += d;
        }
        __syncthreads();

        // ---- G4: dW3 += dOUT^T H2 (columns 32w..) ; G5: dH2 = (dOUT W3) * (H2 > 0), in place -------------------
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = kc * 8 + 4 * hf + j;
                dW3a = mfma(dOs[s * XS + c], H2s[s * HS + col], dW3a);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const f32x4 a = ld4(&dOs[c * XS + kc * 8 + 4 * hf]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma(a[j], w3t[kc * 4 + j], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = acc_row(r, hf) * HS + col;
            const float v = H2s[idx] > 0.0f ? acc[r] : 0.0f;
            db2a += v;
            H2s[idx] = v;
        }
        __syncthreads();

        // ---- G6: dH1 = (dH2 W2) * (H1 > 0) (kept in registers until G7 has read H1) -----------------------------
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.0f;
        {
            f32x4 a = ld4(&H2s[c * HS + 4 * hf]);
            float b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = W2s[(4 * hf + j) * HS + col];
#pragma unroll
            for (int kc = 0; kc < 16; ++kc) {
                f32x4 an = a;
                float bn[4] = {b[0], b[1], b[2], b[3]};
                if (kc + 1 < 16) {
                    an = ld4(&H2s[c * HS + (kc + 1) * 8 + 4 * hf]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j] = W2s[((kc + 1) * 8 + 4 * hf + j) * HS + col];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) dh = mfma(a[j], b[j], dh);
                a = an;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = bn[j];
            }
        }
        // ---- G7: dW2 += dH2^T H1 (rows 32w..) ---------------------------------------------------------------------
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = kc * 8 + 4 * hf + j;
                const float a = H2s[s * HS + col];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) dW2a[cb] = mfma(a, H1s[s * HS + 32 * cb + c], dW2a[cb]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = acc_row(r, hf) * HS + col;
            H1s[idx] = H1s[idx] > 0.0f ? dh[r] : 0.0f;
        }
        // (no barrier: G8 reads only the columns this wave has just written)

        // ---- G8: [dW1 db1] += dH1^T [X 1] (rows 32w..) ------------------------------------------------------------
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = kc * 8 + 4 * hf + j;
                dW1a = mfma(H1s[s * HS + col], Xs[s * XS + c], dW1a);
            }
        }
        __syncthreads();
    }

    // ---- this workgroup's partial gradient, parameter order --------------------------------------------------------
    float *out = p.partial + (size_t)blockIdx.x * p.numel;
    float *oW1 = out, *ob1 = oW1 + H * S, *oW2 = ob1 + H, *ob2 = oW2 + H * H, *oW3 = ob2 + H, *ob3 = oW3 + A * H;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hf);
        const int o = 32 * w + row;
        if (c < S) oW1[o * S + c] = dW1a[r];
        else if (c == S) ob1[o] = dW1a[r];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) oW2[o * H + 32 * cb + c] = dW2a[cb][r];
        if (row < A) oW3[row * H + col] = dW3a[r];
    }
    const float db2t = db2a + __shfl_xor(db2a, 32, 64);
    if (hf == 0) ob2[col] = db2t;
    red[(w * 2 + hf) * 32 + c] = db3a;              // 8 sample groups x 32 logits
    float lw = loss_a;
    for (int off = 32; off >= 1; off >>= 1) lw += __shfl_xor(lw, off, 64);
    if (l == 0) red[256 + w] = lw;
    __syncthreads();
    if (tid < A) {
        float s = 0.0f;
        for (int g = 0; g < 8; ++g) s += red[g * 32 + tid];
        ob3[tid] = s;
    }
    if (tid == 0) p.loss_partial[blockIdx.x] = ((red[256] + red[257]) + red[258]) + red[259];
}

// grad[j] = sum over the workgroups' partials (fixed order); block = 64 columns x 4 row groups.  Block 0 also
// finishes the loss: sign * sum / count.
__global__ __launch_bounds__(256) void grad_finish_kernel(const float *partial, int groups, int numel, float *grad, const float *loss_partial,
                                                          const float *count, float sign, float *loss) {
    __shared__ float sh[256];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    float s = 0.0f;
    if (j < numel) for (int g = rg; g < groups; g += 4) s += partial[(size_t)g * numel + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && j < numel) grad[j] = ((sh[cl] + sh[64 + cl]) + sh[128 + cl]) + sh[192 + cl];
    if (blockIdx.x == 0) {
        __syncthreads();
        float v = 0.0f;
        for (int g = threadIdx.x; g < groups; g += 256) v += loss_partial[g];
        sh[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < 256; ++i) t += sh[i];
            loss[0] = sign * t / count[0];
        }
    }
}

int g_groups = 0;
bool g_lds_ok = false;

}  // namespace

extern "C" {

int fjsp_mlp_train_groups(int32_t n) {
    if (g_groups == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_groups = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles = (n + TS - 1) / TS;
    return tiles < g_groups ? (tiles > 0 ? tiles : 1) : g_groups;
}

int fjsp_mlp_train_pass(int32_t mode, const float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden,
                        int32_t n_out, const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count,
                        float clip_epsilon, float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss,
                        void *stream) {
    if (!d_params || !d_x || !d_aux0 || !d_count || !d_partial || !d_loss_partial || !d_grad || !d_loss || n <= 0 ||
        (mode != 0 && mode != 1) || (mode == 0 && (!d_aux1 || !d_aux2))) {
        fjsp::set_error("fjsp_mlp_train_pass: bad arguments"); return FJSP_E_ARG;
    }
    if (hidden != H || state_size < 1 || state_size > 31 || n_out < 1 || n_out > 32 || (mode == 1 && n_out != 1)) {
        fjsp::set_error("fjsp_mlp_train_pass: supports state_size <= 31, hidden == 128, outputs <= 32 (critic: 1)"); return FJSP_E_UNSUPPORTED;
    }
    if ((reinterpret_cast<uintptr_t>(d_params) & 15) != 0) { fjsp::set_error("fjsp_mlp_train_pass: parameter buffer must be 16-byte aligned"); return FJSP_E_ARG; }
    const int groups = fjsp_mlp_train_groups(n);
    if (n_groups != groups) { fjsp::set_error("fjsp_mlp_train_pass: n_groups must be fjsp_mlp_train_groups(n)"); return FJSP_E_ARG; }
    const size_t lds = (size_t)kLdsFloats * sizeof(float);
    if (!g_lds_ok) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_train_pass_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_train_pass_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            fjsp::set_error("fjsp_mlp_train_pass: cannot raise the dynamic LDS limit"); return FJSP_E_HIP;
        }
        g_lds_ok = true;
    }
    PassArgs a;
    a.params = d_params; a.x = d_x; a.aux0 = d_aux0; a.aux1 = d_aux1; a.aux2 = d_aux2; a.count = d_count;
    a.partial = d_partial; a.loss_partial = d_loss_partial;
    a.n = n; a.S = state_size; a.A = n_out; a.numel = H * state_size + H + H * H + H + n_out * H + n_out; a.clip_eps = clip_epsilon;
    if (mode == 0) hipLaunchKernelGGL(mlp_train_pass_kernel<0>, dim3(groups), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_train_pass_kernel<1>, dim3(groups), dim3(256), lds, (hipStream_t)stream, a);
    hipLaunchKernelGGL(grad_finish_kernel, dim3((a.numel + 63) / 64), dim3(256), 0, (hipStream_t)stream, d_partial, groups, a.numel, d_grad,
                       d_loss_partial, d_count, mode == 0 ? -1.0f : 1.0f, d_loss);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fjsp::set_error(std::string("fjsp_mlp_train_pass: ") + hipGetErrorString(e)); return FJSP_E_HIP; }
    return FJSP_OK;
}

}  // extern "C"
