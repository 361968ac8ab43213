"""Dense layer for tall batches (hundreds of thousands of rollout samples x <= 200 features).

The weight gradient of such a layer is dW[out, in] = sum_s dY[s, out] * X[s, in]: a GEMM whose output is tiny
(128 x 128 = 16 MFMA tiles) and whose reduction dimension is the whole batch.  The library GEMM picked for
that shape runs 16 workgroups on a 256-CU part (rocprofv3: 514 us per call at 229k samples, 40 % of a PPO
round).  `tall_linear` keeps the forward and the input gradient on the library GEMM and computes the weight
gradient as a split-K product: the batch is cut into CHUNKS slices, one batched GEMM produces CHUNKS partial
[out, in] tiles (CHUNKS x 16 workgroups), and a short reduction adds them.  Same f32 arithmetic, different
summation order.
"""
import torch
import torch.nn.functional as F

TALL = 1 << 15          # rows from which the split-K weight gradient pays
CHUNKS = 128


class _TallLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_x = grad_out @ weight if ctx.needs_input_grad[0] else None
        grad_w = grad_b = None
        if ctx.needs_input_grad[1]:
            S = x.shape[0]
            c = CHUNKS
            per = S // c
            head = per * c
            gw = torch.bmm(grad_out[:head].reshape(c, per, -1).transpose(1, 2), x[:head].reshape(c, per, -1)).sum(0)
            if head < S:
                gw = gw + grad_out[head:].t() @ x[head:]
            grad_w = gw
        if ctx.needs_input_grad[2]:
            grad_b = grad_out.sum(0)
        return grad_x, grad_w, grad_b


def tall_linear(x, layer):
    """`layer(x)` for an nn.Linear, with the split-K weight gradient when x is a tall 2-D batch."""
    if x.dim() == 2 and x.shape[0] >= TALL and x.is_cuda:
        return _TallLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


def run_layers(layers, x):
    """Apply a ModuleList of Linear / activation layers, dense layers through `tall_linear`."""
    for layer in layers:
        x = tall_linear(x, layer) if isinstance(layer, torch.nn.Linear) else layer(x)
    return x
