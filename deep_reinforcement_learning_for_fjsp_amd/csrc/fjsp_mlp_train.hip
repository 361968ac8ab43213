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
// Between the products: the ReLU gates of the backward pass are two 16-bit masks per lane (bit r = element
// (acc_row(r), column) was positive), so the epilogues of G5 / G6 only write; the loss stage runs on 8 lanes per
// sample (4 logits per lane, DPP reductions inside the 8-lane group, no LDS round trips); global loads (next tile's
// input rows, the tile's per-sample loss inputs) are issued right after a barrier and land under G2's 64 MFMAs -- a
// barrier waits for every outstanding memory operation of the wave, so a load pending at one stalls for its full
// latency; the critic's single output makes G4 / G5 rank-1 products, done on the vector ALU.
//
// Sums over the samples run in a fixed order (tile order inside a workgroup, then workgroup order in the finish
// kernel): results are deterministic, and differ from the library path by f32 reassociation only.
//
// Measured (MI355X, 163 840 samples, S = 20, A = 24; profiles/r02_*_mlp_pass_*): 249 us (actor) / 209 us (critic) per
// pass = 84-86 TFLOP/s of useful flops (157 peak); a tile takes 24.7 k cycles of which the 272 MFMAs are 17.4 k.
// A two-crew variant (two tiles per workgroup on 2 waves per SIMD, phases offset so that one crew's loss / epilogue
// phases overlap the other's MFMAs) was built and measured 2x slower: at <= 256 registers per wave the compiler
// spills ~150 registers (the six 16-register accumulators plus fragments), and every reload waits for memory.
#include <hip/hip_runtime.h>

#include <algorithm>
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
constexpr int kLdsFloats = H * HS + TS * XS + 2 * TS * HS + 4 * TS * 32 + TS * XS + 8;

struct PassArgs {
    const float *params;      // flat: W1[H][S] b1[H] W2[H][H] b2[H] W3[A][H] b3[A]
    const float *x;           // [n][S]
    const float *aux0;        // actor: actions (f32) | critic: returns
    const float *aux1;        // actor: old log-probabilities
    const float *aux2;        // actor: advantages
    const float *count;       // [1] global sample count (the mean's denominator)
    float *partial;           // [gridDim.x][numel] gradient partials, parameter order
    float *loss_partial;      // [gridDim.x]
    float *vout;              // critic, nullable: the value of every sample as this pass's forward computed it (before the update)
    int n, S, A, numel;
    float clip_eps;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// float reductions over aligned groups of 8 lanes by DPP (no LDS round trip): xor 1, xor 2 inside the quad, then the
// mirrored quad of the 8-lane half row.  Every lane of the group ends with the same bits (each step adds / compares
// the same two values in both lanes: commutative).
#define FJSP_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false))
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, FJSP_DPP_F(v, 0xB1));       // quad_perm [1,0,3,2]
    v = fmaxf(v, FJSP_DPP_F(v, 0x4E));       // quad_perm [2,3,0,1]
    return fmaxf(v, FJSP_DPP_F(v, 0x141));   // row_half_mirror
}
__device__ __forceinline__ float group8_sum(float v) {
    v = v + FJSP_DPP_F(v, 0xB1);
    v = v + FJSP_DPP_F(v, 0x4E);
    return v + FJSP_DPP_F(v, 0x141);
}

#ifdef FJSP_MLP_STAMPS   // diagnostic build (tools/mlp_phase_stamps.py): cycles per phase of the 3rd tile of workgroup 0, wave 0
#define STAMP() do { if (tile == (int)blockIdx.x + 2 * (int)gridDim.x && nstamp < 12) stamp[nstamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP() do { } while (0)
#endif

template <int MODE>   // 0: actor (clipped surrogate on softmax logits), 1: critic (mean squared error on the value)
__global__ __launch_bounds__(256) void mlp_train_pass_kernel(PassArgs p) {
    extern __shared__ float lds[];
#ifdef FJSP_MLP_STAMPS
    const long long k_t0 = __builtin_readcyclecounter(), k_r0 = __builtin_amdgcn_s_memrealtime();
    long long tile_t[24], stamp[12];
    int ntile_t = 0, nstamp = 0;
#endif
    float *W2s = lds;
    float *Xs = W2s + H * HS;
    float *H1s = Xs + TS * XS;
    float *H2s = H1s + TS * HS;
    float *OUTp = H2s + TS * HS;
    float *dOs = OUTp + 4 * TS * 32;
    float *red = dOs + TS * XS;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, c = l & 31, hf = l >> 5;
    const int ls = tid >> 3, lg = tid & 7;      // loss stage: sample ls, logits 4 lg .. 4 lg + 3
    const int S = p.S, A = p.A, n = p.n;
    const float *W1 = p.params, *b1 = W1 + H * S, *W2 = b1 + H, *b2 = W2 + H * H, *W3 = b2 + H, *b3 = W3 + A * H;
    const int col = 32 * w + c;               // the output column / weight row this lane serves

    for (int i = tid; i < H * H / 4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(W2)[i];
        *reinterpret_cast<f32x4 *>(&W2s[(i >> 5) * HS + (i & 31) * 4]) = v;
    }
    // X image: column S = 1 (bias), columns above it 0; columns 0..S-1 rewritten per tile
    for (int i = tid; i < TS * 32; i += 256) Xs[(i >> 5) * XS + (i & 31)] = ((i & 31) == S) ? 1.0f : 0.0f;

    float w1f[16], w3f[16], w3t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = (q >> 2) * 8 + 4 * hf + (q & 3);
        // (loads of clamped indices times a 0/1 factor: branch-free, so the 48 loads are in flight together)
        const int k1 = k < S ? col * S + k : (k == S ? H * S + col : -1);     // (b1 follows W1 in the flat buffer)
        w1f[q] = W1[k1 >= 0 ? k1 : 0] * (k1 >= 0 ? 1.0f : 0.0f);              // B[k][o] of G1
        w3f[q] = W3[(c < A ? c : 0) * H + 32 * w + k] * (c < A ? 1.0f : 0.0f);   // B[k = h][a] of G3 (K quarter w)
        w3t[q] = MODE == 0 ? W3[(k < A ? k : 0) * H + col] * (k < A ? 1.0f : 0.0f) : 0.0f;   // B[k = a][h] of G5
    }
    const float b2c = b2[col];
    const float b3c = (w == 0 && c < A) ? b3[c] : 0.0f;
    const float w3c = W3[col];                 // critic: the value head's weight of this column
    const float inv_count = 1.0f / p.count[0];

    f32x16 dW2a[4], dW1a, dW3a;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW1a[r] = 0.0f; dW3a[r] = 0.0f; dW2a[0][r] = 0.0f; dW2a[1][r] = 0.0f; dW2a[2][r] = 0.0f; dW2a[3][r] = 0.0f; }
    float db2a = 0.0f, loss_a = 0.0f, dW3c = 0.0f;
    float db3a[4] = {0.0f, 0.0f, 0.0f, 0.0f};

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
            // Unconditional load of a clamped index, no select: a load under a branch is followed by a full wait.  Lanes
            // without a share never store their value; rows beyond n get finite data of the last rows, and their
            // d logits are 0, so they add exact zeros to every gradient.
            const size_t idx = (size_t)tile * TS * S + tid + 256 * q, last = (size_t)n * S - 1;
            xr[q] = p.x[idx < last ? idx : last];
        }
    };
    load_x(blockIdx.x);
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef FJSP_MLP_STAMPS
        if (ntile_t < 24) tile_t[ntile_t++] = __builtin_readcyclecounter();
#endif
        STAMP();
#pragma unroll
        for (int q = 0; q < 4; ++q) if (xdst[q] >= 0) Xs[xdst[q]] = xr[q];
        __syncthreads();
        STAMP();

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
        unsigned mask1 = 0, mask2 = 0;            // bit r: H1 / H2 element (acc_row(r), col) is positive (the ReLU gates of the backward pass)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mask1 |= (acc[r] > 0.0f ? 1u : 0u) << r;
            H1s[acc_row(r, hf) * HS + col] = fmaxf(acc[r], 0.0f);
        }
        __syncthreads();
        STAMP();
        // Global loads are issued right after a barrier and land under G2's 64 MFMAs: a barrier waits for every
        // outstanding memory operation of the wave, a load pending there is a stall of its full latency.
        const int gs = tile * TS + ls;
        const bool valid = gs < n;
        const int gsc = valid ? gs : n - 1;
        const float m0 = p.aux0[gsc];
        float m1 = 0.0f, m2 = 0.0f;
        if (MODE == 0) { m1 = p.aux1[gsc]; m2 = p.aux2[gsc]; }
        load_x(tile + gridDim.x);

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
        for (int r = 0; r < 16; ++r) {
            mask2 |= (acc[r] > 0.0f ? 1u : 0u) << r;
            H2s[acc_row(r, hf) * HS + col] = fmaxf(acc[r], 0.0f);
        }
        // (no barrier: G3 reads only the columns this wave has just written)
        STAMP();

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
        STAMP();

        // ---- loss and d loss / d logits: 8 lanes per sample, 4 logits per lane ------------------------------------
        if (MODE == 0) {
            const float *zp = &OUTp[ls * 32 + 4 * lg];
            const f32x4 z4 = ((ld4(zp) + ld4(zp + TS * 32)) + ld4(zp + 2 * TS * 32)) + ld4(zp + 3 * TS * 32);
            const int act = (int)m0;
            const float z_act = ((OUTp[ls * 32 + act] + OUTp[TS * 32 + ls * 32 + act]) + OUTp[2 * TS * 32 + ls * 32 + act]) + OUTp[3 * TS * 32 + ls * 32 + act];
            float zz[4], e[4];
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) { zz[i] = (4 * lg + i < A) ? z4[i] : -INFINITY; m = fmaxf(m, zz[i]); }
            m = group8_max(m);
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { e[i] = (4 * lg + i < A) ? expf(zz[i] - m) : 0.0f; sum += e[i]; }
            sum = group8_sum(sum);
            f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
            if (valid) {
                const float lse = m + logf(sum);
                const float new_lp = z_act - lse;                                   // log_softmax(...).gather(action), :327-328
                const float ratio = expf(new_lp) / (expf(m1) + 1e-8f);               // :330-333
                const float lo = 1.0f - p.clip_eps, hi = 1.0f + p.clip_eps;
                const float clipped = fminf(fmaxf(ratio, lo), hi);
                const float s1 = m2 * ratio, s2 = m2 * clipped;                      // :344-350
                const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;     // autograd: clamp passes the gradient on [lo, hi],
                float g;                                                             // minimum splits a tie in half
                if (s1 < s2) g = m2;
                else if (s2 < s1) g = m2 * inside;
                else g = 0.5f * m2 + 0.5f * m2 * inside;
                const float cf = -g * ratio * inv_count;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = (4 * lg + i < A) ? cf * ((4 * lg + i == act ? 1.0f : 0.0f) - e[i] / sum) : 0.0f;
                if (lg == 0) loss_a += fminf(s1, s2);
            }
            *reinterpret_cast<f32x4 *>(&dOs[ls * XS + 4 * lg]) = d;
#pragma unroll
            for (int i = 0; i < 4; ++i) db3a[i] += d[i];
        } else {
            if (lg == 0) {
                const float z = ((OUTp[ls * 32] + OUTp[TS * 32 + ls * 32]) + OUTp[2 * TS * 32 + ls * 32]) + OUTp[3 * TS * 32 + ls * 32];
                float d = 0.0f;
                if (valid && p.vout) p.vout[gs] = z;                                    // V(s) for the advantages (:263)
                if (valid) {
                    const float dv = z - m0;                                            // F.mse_loss, :317-318
                    d = 2.0f * dv * inv_count;
                    loss_a += dv * dv;
                }
                dOs[ls] = d;                                                            // critic: d value [32], contiguous
                db3a[0] += d;
            }
        }
        __syncthreads();
        STAMP();

        if (MODE == 0) {
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
                const float v = ((mask2 >> r) & 1u) ? acc[r] : 0.0f;
                db2a += v;
                H2s[acc_row(r, hf) * HS + col] = v;
            }
        } else {
            // critic: one output -> rank-1 products on the vector ALU: dW3[col] += sum_s d[s] H2[s][col]; dH2 = d[s] W3[col] gated
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 d4 = ld4(&dOs[8 * q + 4 * hf]);             // rows acc_row(4q .. 4q+3, hf) = 8q + 4hf + (0..3)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i, idx = acc_row(r, hf) * HS + col;
                    dW3c = fmaf(d4[i], H2s[idx], dW3c);
                    const float v = ((mask2 >> r) & 1u) ? d4[i] * w3c : 0.0f;
                    db2a += v;
                    H2s[idx] = v;
                }
            }
        }
        __syncthreads();
        STAMP();

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
        STAMP();
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
        STAMP();
#pragma unroll
        for (int r = 0; r < 16; ++r) H1s[acc_row(r, hf) * HS + col] = ((mask1 >> r) & 1u) ? dh[r] : 0.0f;
        // (no barrier: G8 reads only the columns this wave has just written)
        STAMP();

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
        STAMP();
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
        if (MODE == 0 && row < A) oW3[row * H + col] = dW3a[r];
    }
    const float db2t = db2a + __shfl_xor(db2a, 32, 64);
    if (hf == 0) ob2[col] = db2t;
    if (MODE == 1) {
        const float t = dW3c + __shfl_xor(dW3c, 32, 64);
        if (hf == 0) oW3[col] = t;
    }
    float *red4 = H1s;                               // (the activation images are dead) [32 samples][32 logits]
#pragma unroll
    for (int i = 0; i < 4; ++i) red4[ls * 32 + 4 * lg + i] = db3a[i];
    float lw = loss_a;
    for (int off = 32; off >= 1; off >>= 1) lw += __shfl_xor(lw, off, 64);
    if (l == 0) red[w] = lw;
    __syncthreads();
    if (tid < A) {
        float s = 0.0f;
        for (int g = 0; g < TS; ++g) s += red4[g * 32 + tid];
        ob3[tid] = s;
    }
    if (tid == 0) p.loss_partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
#ifdef FJSP_MLP_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {             // (the loss partials of workgroups 16.. are overwritten: diagnostic build)
        for (int i = 0; i + 1 < nstamp; ++i) p.loss_partial[16 + i] = (float)(stamp[i + 1] - stamp[i]);
        const long long k_t1 = __builtin_readcyclecounter(), k_r1 = __builtin_amdgcn_s_memrealtime();
        p.loss_partial[32] = (float)(k_t1 - k_t0);                 // whole kernel, shader cycles
        p.loss_partial[33] = (float)(k_r1 - k_r0);                 // whole kernel, 100 MHz ticks
        p.loss_partial[34] = (float)(tile_t[0] - k_t0);            // prologue
        for (int i = 0; i + 1 < ntile_t; ++i) p.loss_partial[36 + i] = (float)(tile_t[i + 1] - tile_t[i]);
        p.loss_partial[35] = (float)(k_t1 - tile_t[ntile_t - 1]);  // last tile + epilogue
    }
#endif
}

// grad[j] = sum over the workgroups' partials (fixed order); block = 64 columns x 8 row groups, the row loop unrolled
// so that a thread's loads are all in flight together.  Block 0 also finishes the loss: sign * sum / count.
// With sumsq_partial != nullptr (the single-process optimiser step that follows in the same stream): also
// sumsq_partial[block] = sum of the squares of the block's 64 gradient entries (clip_grad_norm_'s total norm), and
// block 0 advances the optimiser's step count.
__global__ __launch_bounds__(512) void grad_finish_kernel(const float *partial, int groups, int numel, float *grad, const float *loss_partial,
                                                          const float *count, float sign, float *loss, float *sumsq_partial, float *step) {
    __shared__ float sh[512];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    float s = 0.0f;
    if (j < numel) {
        int g = rg;
        for (; g + 56 < groups; g += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(g + 8 * u) * numel + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < groups; g += 8) s += partial[(size_t)g * numel + j];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && j < numel) {
        float t = sh[cl];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sh[64 * q + cl];
        grad[j] = t;
        s = t * t;
    } else {
        s = 0.0f;
    }
    if (sumsq_partial) {                              // (rg == 0 is wave 0: its 64 lanes hold the block's squares)
        if (rg == 0) {
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (cl == 0) sumsq_partial[blockIdx.x] = s;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) step[0] += 1.0f;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        float v = 0.0f;
        for (int g = threadIdx.x; g < groups; g += 512) v += loss_partial[g];
        sh[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < 512; ++i) t += sh[i];
            loss[0] = sign * t / count[0];
        }
    }
}

// clip_grad_norm_ (coefficient from the finish kernel's partial sums of squares) + Adam (torch.optim.Adam, no weight
// decay, no amsgrad); *step is the count AFTER this step (advanced by the finish kernel).  Same arithmetic as
// adam_clip_kernel of fjsp_ppo.hip.
__global__ __launch_bounds__(256) void adam_apply_kernel(float *p, const float *g, float *m, float *v, int n, const float *sumsq_partial, int nparts,
                                                         float max_norm, float lr, float beta1, float beta2, float eps, const float *step) {
    __shared__ float sh[256];
    __shared__ float coef_s, bc1_s, bc2_s;
    float ss = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += 256) ss += sumsq_partial[i];
    sh[threadIdx.x] = ss;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = sqrtf(sh[0]);
        coef_s = max_norm > 0.0f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;
        const float t = step[0];
        bc1_s = 1.0f - powf(beta1, t);
        bc2_s = 1.0f - powf(beta2, t);
    }
    __syncthreads();
    const float coef = coef_s, bc1 = bc1_s, bc2 = bc2_s;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// per device: the workgroup count (= its CU count) and whether the kernels' dynamic-LDS limit has been raised there
constexpr int kMaxDevices = 64;
int g_groups[kMaxDevices] = {};
bool g_lds_ok[kMaxDevices] = {};

int groups_of_device(int dev) {
    if (dev < 0 || dev >= kMaxDevices) return 256;
    if (g_groups[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_groups[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_groups[dev];
}

// the device a buffer lives on (-1: not a device pointer)
int device_of(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.type == hipMemoryTypeDevice ? at.device : -1;
}

struct OnDevice {          // launches below go to the parameters' device, whatever the caller's current device is
    int prev = -1;
    explicit OnDevice(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~OnDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

static int groups_for(int32_t n, int dev) {
    const int g = groups_of_device(dev), tiles = (n + TS - 1) / TS;
    return tiles < g ? (tiles > 0 ? tiles : 1) : g;
}

int fjsp_mlp_train_groups(int32_t n) {          // (for the caller's current device: where it allocates the buffers)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    return groups_for(n, dev);
}

namespace {
int train_pass_launch(int32_t mode, const float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden, int32_t n_out,
                      const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count, float clip_epsilon, float *d_partial,
                      int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss, float *d_sumsq_partial, float *d_step, void *stream,
                      int *numel_out, float *d_values_out = nullptr) {
    if (!d_params || !d_x || !d_aux0 || !d_count || !d_partial || !d_loss_partial || !d_grad || !d_loss || n <= 0 ||
        (mode != 0 && mode != 1) || (mode == 0 && (!d_aux1 || !d_aux2))) {
        fjsp::set_error("fjsp_mlp_train_pass: bad arguments"); return FJSP_E_ARG;
    }
    if (hidden != H || state_size < 1 || state_size > 31 || n_out < 1 || n_out > 32 || (mode == 1 && n_out != 1)) {
        fjsp::set_error("fjsp_mlp_train_pass: supports state_size <= 31, hidden == 128, outputs <= 32 (critic: 1)"); return FJSP_E_UNSUPPORTED;
    }
    if ((reinterpret_cast<uintptr_t>(d_params) & 15) != 0) { fjsp::set_error("fjsp_mlp_train_pass: parameter buffer must be 16-byte aligned"); return FJSP_E_ARG; }
    const int dev = device_of(d_params);
    if (dev < 0 || dev >= kMaxDevices) { fjsp::set_error("fjsp_mlp_train_pass: the parameter buffer is not device memory"); return FJSP_E_ARG; }
    OnDevice on(dev);
    const int groups = groups_for(n, dev);
    if (n_groups != groups) { fjsp::set_error("fjsp_mlp_train_pass: n_groups must be fjsp_mlp_train_groups(n) taken on the parameters' device"); return FJSP_E_ARG; }
    const size_t lds = (size_t)kLdsFloats * sizeof(float);
    if (!g_lds_ok[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_train_pass_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_train_pass_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            fjsp::set_error("fjsp_mlp_train_pass: cannot raise the dynamic LDS limit"); return FJSP_E_HIP;
        }
        g_lds_ok[dev] = true;
    }
    PassArgs a;
    a.params = d_params; a.x = d_x; a.aux0 = d_aux0; a.aux1 = d_aux1; a.aux2 = d_aux2; a.count = d_count;
    a.partial = d_partial; a.loss_partial = d_loss_partial; a.vout = mode == 1 ? d_values_out : nullptr;
    a.n = n; a.S = state_size; a.A = n_out; a.numel = H * state_size + H + H * H + H + n_out * H + n_out; a.clip_eps = clip_epsilon;
    if (mode == 0) hipLaunchKernelGGL(mlp_train_pass_kernel<0>, dim3(groups), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_train_pass_kernel<1>, dim3(groups), dim3(256), lds, (hipStream_t)stream, a);
    hipLaunchKernelGGL(grad_finish_kernel, dim3((a.numel + 63) / 64), dim3(512), 0, (hipStream_t)stream, d_partial, groups, a.numel, d_grad,
                       d_loss_partial, d_count, mode == 0 ? -1.0f : 1.0f, d_loss, d_sumsq_partial, d_step);
    *numel_out = a.numel;
    return FJSP_OK;
}
}  // namespace

int fjsp_mlp_train_pass(int32_t mode, const float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden,
                        int32_t n_out, const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count,
                        float clip_epsilon, float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss,
                        void *stream) {
    int numel = 0;
    const int rc = train_pass_launch(mode, d_params, d_x, n, state_size, hidden, n_out, d_aux0, d_aux1, d_aux2, d_count, clip_epsilon, d_partial,
                                     n_groups, d_loss_partial, d_grad, d_loss, nullptr, nullptr, stream, &numel);
    if (rc != FJSP_OK) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fjsp::set_error(std::string("fjsp_mlp_train_pass: ") + hipGetErrorString(e)); return FJSP_E_HIP; }
    return FJSP_OK;
}

static int train_step_impl(int32_t mode, float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden, int32_t n_out,
                           const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count, float clip_epsilon,
                           float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss, float *d_exp_avg,
                           float *d_exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float *d_step,
                           float *d_sumsq_partial, void *stream, float *d_values_out) {
    if (!d_exp_avg || !d_exp_avg_sq || !d_step || !d_sumsq_partial) { fjsp::set_error("fjsp_mlp_train_step: bad arguments"); return FJSP_E_ARG; }
    int numel = 0;
    const int rc = train_pass_launch(mode, d_params, d_x, n, state_size, hidden, n_out, d_aux0, d_aux1, d_aux2, d_count, clip_epsilon, d_partial,
                                     n_groups, d_loss_partial, d_grad, d_loss, d_sumsq_partial, d_step, stream, &numel, d_values_out);
    if (rc != FJSP_OK) return rc;
    const int blocks = std::min(256, (numel + 255) / 256);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_params, d_grad, d_exp_avg, d_exp_avg_sq, numel,
                       d_sumsq_partial, (numel + 63) / 64, max_norm, lr, beta1, beta2, eps, d_step);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fjsp::set_error(std::string("fjsp_mlp_train_step: ") + hipGetErrorString(e)); return FJSP_E_HIP; }
    return FJSP_OK;
}

int fjsp_mlp_train_step(int32_t mode, float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden, int32_t n_out,
                        const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count, float clip_epsilon,
                        float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss, float *d_exp_avg,
                        float *d_exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float *d_step,
                        float *d_sumsq_partial, void *stream) {
    return train_step_impl(mode, d_params, d_x, n, state_size, hidden, n_out, d_aux0, d_aux1, d_aux2, d_count, clip_epsilon, d_partial, n_groups,
                           d_loss_partial, d_grad, d_loss, d_exp_avg, d_exp_avg_sq, max_norm, lr, beta1, beta2, eps, d_step, d_sumsq_partial,
                           stream, nullptr);
}

int fjsp_mlp_train_step_values(int32_t mode, float *d_params, const float *d_x, int32_t n, int32_t state_size, int32_t hidden, int32_t n_out,
                               const float *d_aux0, const float *d_aux1, const float *d_aux2, const float *d_count, float clip_epsilon,
                               float *d_partial, int32_t n_groups, float *d_loss_partial, float *d_grad, float *d_loss, float *d_exp_avg,
                               float *d_exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float *d_step,
                               float *d_sumsq_partial, float *d_values_out, void *stream) {
    if (mode != 1 || !d_values_out) { fjsp::set_error("fjsp_mlp_train_step_values: the critic pass (mode 1) and an output buffer are required"); return FJSP_E_ARG; }
    return train_step_impl(mode, d_params, d_x, n, state_size, hidden, n_out, d_aux0, d_aux1, d_aux2, d_count, clip_epsilon, d_partial, n_groups,
                           d_loss_partial, d_grad, d_loss, d_exp_avg, d_exp_avg_sq, max_norm, lr, beta1, beta2, eps, d_step, d_sumsq_partial,
                           stream, d_values_out);
}

}  // extern "C"
