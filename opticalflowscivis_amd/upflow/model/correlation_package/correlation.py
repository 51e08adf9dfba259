"""Drop-in for UPFlow/model/correlation_package/correlation.py: `CorrelationFunction` and the
`Correlation` module, bound to the HIP `correlation_cuda` mirror."""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import correlation_cuda


class CorrelationFunction(Function):
    """correlation.py:6-45 -- same defaults, same saved state, 8 gradients (6 of them None)."""

    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1,
                stride2=2, corr_multiply=1):
        ctx.cfg = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        ctx.save_for_backward(input1, input2)
        with torch.cuda.device_of(input1):
            rbot1, rbot2, output = input1.new(), input2.new(), input1.new()
            correlation_cuda.forward(input1, input2, rbot1, rbot2, output, *ctx.cfg)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        with torch.cuda.device_of(input1):
            rbot1, rbot2 = input1.new(), input2.new()
            grad_input1, grad_input2 = input1.new(), input2.new()
            correlation_cuda.backward(input1, input2, rbot1, rbot2, grad_output, grad_input1,
                                      grad_input2, *ctx.cfg)
        return grad_input1, grad_input2, None, None, None, None, None, None


class Correlation(Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2,
                 corr_multiply=1):
        super().__init__()
        self.cfg = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)

    def forward(self, input1, input2):
        return CorrelationFunction.apply(input1, input2, *self.cfg)
