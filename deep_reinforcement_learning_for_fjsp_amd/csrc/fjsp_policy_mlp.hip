// Action sampling of the HMPSAC policies in ONE launch per (task, machine) pair
// (agents/HMPSAC/SAC_Discrete.py:277-284 pick_lower_action, :248-254 pick_action; A3C_v5.1.py:35-75 the two policy networks).
//
// The reference evaluates, per environment step, TaskPolicyNet(state) -> Categorical sample a_t, then
// MachinePolicyNet(cat(state, a_t)) -> Categorical sample a_m; batched over 4096 environments that is, through the library,
// 4 GEMMs + bias / ReLU / softmax / multinomial kernels per network -- ~18 launches of ~4 us for 0.7 GFLOP, seven networks
// per controller step.  Here a workgroup takes 16 environments through both networks: activations in LDS (two [256][16]
// f32 buffers, the 16 rows of a feature contiguous), thread j owns output neuron j of a layer for the 16 rows (16
// accumulators; per input one weight load -- the weights are passed TRANSPOSED, [in][out], so the threads of a wave read
// consecutive words: with torch's [out][in] every lane streams its own row, a wave touches 64 cache lines per load and the
// launch moves 8x the weights through L2 -- and four broadcast float4 reads of the inputs), then one lane per row does the
// softmax (expf(x - max) / sum, f32 as torch) and draws the action by inverse CDF from a counter-based splitmix64 stream
// (seed, row, per-row draw counter kept in device memory, so a launch replayed from a HIP graph keeps drawing fresh numbers).
//
// Supported shapes: Linear-ReLU-...-Linear with 1..6 linear layers, every width <= 256, <= 64 outputs.  Not a training path:
// forward + sample only (the networks' updates stay where they were).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fjsp_amd.h"
#include "fjsp_common.h"
#include "fjsp_host.h"

namespace {

constexpr int kRows = 16;           // environments per workgroup
constexpr int kWidth = 256;         // widest layer; threads per workgroup
constexpr int kMaxLayers = 6;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct Mlp {
    const float *W[kMaxLayers];     // [in][out] row-major: the TRANSPOSE of torch.nn.Linear.weight
    const float *B[kMaxLayers];
    int dims[kMaxLayers + 1];       // dims[0] = inputs, dims[n] = outputs
    int n;                          // linear layers (0 = network absent)
};

struct PairArgs {
    Mlp task, machine;
    const double *state;            // f64[rows][S]
    int rows, S;
    unsigned long long seed;
    unsigned int *draws;            // u32[rows]: draws made so far by each row's stream
    long long *a_task, *a_machine;  // i64[rows]
    float *p_task, *p_machine;      // nullable: f32[rows][outputs], the probabilities the samples were drawn from
    unsigned char *pair;            // nullable: u8[rows][2] = (a_task, a_machine) in the environment's action encoding ...
    const long long *select;        // ... written only where select[row] == which (nullable: everywhere)
    int which;
};

// One network over the workgroup's rows: activations as [feature][kRows]; in = buf_a (first dims[0] features valid), the
// logits end up in the buffer the function returns.
__device__ float *mlp_forward(const Mlp &m, float *buf_a, float *buf_b, int tid) {
    float *in = buf_a, *out = buf_b;
    for (int l = 0; l < m.n; ++l) {
        const int ni = m.dims[l], no = m.dims[l + 1];
        if (tid < no) {
            // 16 accumulators as 8 pairs: v_pk_fma_f32 does two f32 FMAs per lane and instruction (the rate the chip's f32
            // vector peak is quoted at); one weight (broadcast to both halves) times two rows' inputs.  (Tried: every wave
            // taking four rows through all neurons, four neurons per lane -- a quarter of the LDS reads, but every wave then
            // streams all the weights: 178 us against 71.)
            v2f acc[kRows / 2];
            const float bias = m.B[l][tid];
#pragma unroll
            for (int r = 0; r < kRows / 2; ++r) acc[r] = v2f{bias, bias};
            const float *w = m.W[l] + tid;                       // column tid of the transposed weights: stride no
            int i = 0;
            for (; i + 4 <= ni; i += 4) {                        // (four weight loads in flight)
                const float w0 = w[(size_t)i * no], w1 = w[(size_t)(i + 1) * no], w2 = w[(size_t)(i + 2) * no], w3 = w[(size_t)(i + 3) * no];
                const v2f W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3};
#pragma unroll
                for (int q = 0; q < kRows / 4; ++q) {
                    const v4f x0 = *reinterpret_cast<const v4f *>(in + (i + 0) * kRows + 4 * q), x1 = *reinterpret_cast<const v4f *>(in + (i + 1) * kRows + 4 * q);
                    const v4f x2 = *reinterpret_cast<const v4f *>(in + (i + 2) * kRows + 4 * q), x3 = *reinterpret_cast<const v4f *>(in + (i + 3) * kRows + 4 * q);
                    acc[2 * q] = __builtin_elementwise_fma(W0, x0.xy, acc[2 * q]); acc[2 * q + 1] = __builtin_elementwise_fma(W0, x0.zw, acc[2 * q + 1]);
                    acc[2 * q] = __builtin_elementwise_fma(W1, x1.xy, acc[2 * q]); acc[2 * q + 1] = __builtin_elementwise_fma(W1, x1.zw, acc[2 * q + 1]);
                    acc[2 * q] = __builtin_elementwise_fma(W2, x2.xy, acc[2 * q]); acc[2 * q + 1] = __builtin_elementwise_fma(W2, x2.zw, acc[2 * q + 1]);
                    acc[2 * q] = __builtin_elementwise_fma(W3, x3.xy, acc[2 * q]); acc[2 * q + 1] = __builtin_elementwise_fma(W3, x3.zw, acc[2 * q + 1]);
                }
            }
            for (; i < ni; ++i) {
                const float wv = w[(size_t)i * no];
                const v2f Wv = {wv, wv};
#pragma unroll
                for (int r = 0; r < kRows / 2; ++r) acc[r] = __builtin_elementwise_fma(Wv, *reinterpret_cast<const v2f *>(in + i * kRows + 2 * r), acc[r]);
            }
            const bool relu = l + 1 < m.n;
#pragma unroll
            for (int r = 0; r < kRows / 2; ++r) {
                out[tid * kRows + 2 * r] = relu ? fmaxf(acc[r].x, 0.0f) : acc[r].x;
                out[tid * kRows + 2 * r + 1] = relu ? fmaxf(acc[r].y, 0.0f) : acc[r].y;
            }
        }
        __syncthreads();
        float *t = in; in = out; out = t;
    }
    return in;
}

// softmax of a row's logits (in place) and one draw from it; lane r of wave 0 handles row r
__device__ int softmax_sample(float *logits, int no, unsigned long long seed, unsigned row, unsigned draw, float *probs_out) {
    // (logits of this row: element a at logits[a * kRows])
    float mx = logits[0];
    for (int a = 1; a < no; ++a) mx = fmaxf(mx, logits[a * kRows]);
    float sum = 0.0f;
    for (int a = 0; a < no; ++a) { const float e = expf(logits[a * kRows] - mx); logits[a * kRows] = e; sum += e; }
    const float inv = 1.0f / sum;
    // u in [0, 1): 24 random bits, the stream of this row (fjsp_common.h splitmix64)
    const unsigned long long h = fjsp::splitmix64(seed + (unsigned long long)row * 0x9E3779B97F4A7C15ull + (unsigned long long)draw * 1000003ull);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    int pick = no - 1;
    float c = 0.0f;
    bool found = false;
    for (int a = 0; a < no; ++a) {
        const float p = logits[a * kRows] * inv;
        if (probs_out) probs_out[a] = p;
        c += p;
        if (!found && u < c) { pick = a; found = true; }
    }
    return pick;
}

__global__ __launch_bounds__(kWidth) void policy_pair_kernel(PairArgs a) {
    __shared__ __attribute__((aligned(16))) float buf0[kRows * kWidth];
    __shared__ __attribute__((aligned(16))) float buf1[kRows * kWidth];
    __shared__ int s_task[kRows];
    const int tid = (int)threadIdx.x;
    const int row0 = (int)blockIdx.x * kRows;
    // the rows' states as f32 (the reference's state.float())
    for (int q = tid; q < kRows * a.S; q += kWidth) {
        const int r = q / a.S, c = q - r * a.S;
        const int row = min(row0 + r, a.rows - 1);
        buf0[c * kRows + r] = (float)a.state[(size_t)row * a.S + c];
    }
    __syncthreads();
    float *logits = mlp_forward(a.task, buf0, buf1, tid);
    const int no_t = a.task.dims[a.task.n];
    if (tid < kRows) {
        const int row = row0 + tid;
        int pick = 0;
        if (row < a.rows) {
            const unsigned d = a.draws[row];
            pick = softmax_sample(logits + tid, no_t, a.seed, (unsigned)row, d, a.p_task ? a.p_task + (size_t)row * no_t : nullptr);
            a.a_task[row] = pick;
            a.draws[row] = d + (a.machine.n ? 2u : 1u);
        }
        s_task[tid] = pick;
    }
    __syncthreads();
    if (!a.machine.n) return;
    // machine network on cat(state, a_t)
    for (int q = tid; q < kRows * (a.S + 1); q += kWidth) {
        const int r = q / (a.S + 1), c = q - r * (a.S + 1);
        const int row = min(row0 + r, a.rows - 1);
        buf0[c * kRows + r] = c < a.S ? (float)a.state[(size_t)row * a.S + c] : (float)s_task[r];
    }
    __syncthreads();
    logits = mlp_forward(a.machine, buf0, buf1, tid);
    const int no_m = a.machine.dims[a.machine.n];
    if (tid < kRows) {
        const int row = row0 + tid;
        if (row < a.rows) {
            const unsigned d = a.draws[row] - 1u;
            const int pick_m = softmax_sample(logits + tid, no_m, a.seed, (unsigned)row, d, a.p_machine ? a.p_machine + (size_t)row * no_m : nullptr);
            a.a_machine[row] = pick_m;
            if (a.pair && (!a.select || a.select[row] == (long long)a.which)) {
                a.pair[2 * row] = (unsigned char)s_task[tid];
                a.pair[2 * row + 1] = (unsigned char)pick_m;
            }
        }
    }
}

int fill(Mlp &m, int n_layers, const int32_t *dims, const float *const *weights, const float *const *biases, int want_in, const char *what) {
    m.n = 0;
    if (n_layers == 0) return FJSP_OK;
    if (n_layers < 1 || n_layers > kMaxLayers || !dims || !weights || !biases) { fjsp::set_error(std::string("fjsp_policy_pair_sample: bad ") + what + " network description"); return FJSP_E_ARG; }
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1 || dims[l] > kWidth) { fjsp::set_error(std::string("fjsp_policy_pair_sample: ") + what + " layer widths must be 1..256"); return FJSP_E_UNSUPPORTED; }
    if (dims[n_layers] > 64) { fjsp::set_error(std::string("fjsp_policy_pair_sample: ") + what + " network has more than 64 outputs"); return FJSP_E_UNSUPPORTED; }
    if (dims[0] != want_in) { fjsp::set_error(std::string("fjsp_policy_pair_sample: ") + what + " network's input width does not match the state"); return FJSP_E_ARG; }
    for (int l = 0; l < n_layers; ++l) {
        if (!weights[l] || !biases[l]) { fjsp::set_error("fjsp_policy_pair_sample: null parameter pointer"); return FJSP_E_ARG; }
        m.W[l] = weights[l]; m.B[l] = biases[l];
    }
    for (int l = 0; l <= n_layers; ++l) m.dims[l] = dims[l];
    m.n = n_layers;
    return FJSP_OK;
}

}  // namespace

extern "C" int fjsp_policy_pair_sample(int32_t task_layers, const int32_t *task_dims, const float *const *task_w, const float *const *task_b,
                                       int32_t machine_layers, const int32_t *machine_dims, const float *const *machine_w,
                                       const float *const *machine_b, const double *d_state, int32_t rows, int32_t state_size, uint64_t seed,
                                       uint32_t *d_draws, int64_t *d_a_task, int64_t *d_a_machine, float *d_p_task, float *d_p_machine,
                                       uint8_t *d_pair, const int64_t *d_select, int32_t which, void *stream) {
    if (!d_state || !d_draws || !d_a_task || rows <= 0 || state_size < 1 || state_size >= kWidth || (machine_layers && !d_a_machine)) {
        fjsp::set_error("fjsp_policy_pair_sample: bad arguments"); return FJSP_E_ARG;
    }
    PairArgs a{};
    int rc = fill(a.task, task_layers, task_dims, task_w, task_b, state_size, "task");
    if (rc != FJSP_OK) return rc;
    if (task_layers < 1) { fjsp::set_error("fjsp_policy_pair_sample: the task network is required"); return FJSP_E_ARG; }
    rc = fill(a.machine, machine_layers, machine_dims, machine_w, machine_b, state_size + 1, "machine");
    if (rc != FJSP_OK) return rc;
    a.state = d_state; a.rows = rows; a.S = state_size; a.seed = seed; a.draws = d_draws;
    a.a_task = reinterpret_cast<long long *>(d_a_task); a.a_machine = reinterpret_cast<long long *>(d_a_machine);
    a.p_task = d_p_task; a.p_machine = d_p_machine;
    if (d_pair && !machine_layers) { fjsp::set_error("fjsp_policy_pair_sample: d_pair needs the machine network"); return FJSP_E_ARG; }
    a.pair = d_pair; a.select = reinterpret_cast<const long long *>(d_select); a.which = which;
    hipLaunchKernelGGL(policy_pair_kernel, dim3((unsigned)((rows + kRows - 1) / kRows)), dim3(kWidth), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fjsp::set_error(std::string("fjsp_policy_pair_sample: ") + hipGetErrorString(e)); return FJSP_E_HIP; }
    return FJSP_OK;
}
