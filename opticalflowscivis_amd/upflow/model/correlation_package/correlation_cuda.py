"""Drop-in for the `correlation_cuda` torch extension the reference binds at
UPFlow/model/correlation_package/correlation.py:4 and builds from sources that are not in its
tree (setup.py:21-32: correlation_cuda.cc, correlation_cuda_kernel.cu).

Same two callables, same positional signature, same in-place contract: the caller passes EMPTY
tensors (`input1.new()`), the callee resizes and fills them; the return value is ignored;
work is enqueued on the current stream of input1's device.  `rbot1`/`rbot2` (the CUDA
implementation's padded channels-last scratch copies) are accepted and left untouched: the HIP
kernel stages its search window in LDS instead.
"""
from .... import ops


def _check(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply):
    # the only configuration the reference ever uses (upflow.py:649,652) -- the same restriction its
    # own PyTorch stand-in asserts (pytorch_correlation.py:17-18)
    if not (pad_size == max_displacement and kernel_size == 1 and stride1 == 1 and stride2 == 1
            and corr_multiply == 1):
        raise ValueError("correlation_cuda (HIP): only pad_size == max_displacement, kernel_size = 1, "
                         "stride1 = stride2 = 1, corr_multiply = 1 is supported; got %r" %
                         ((pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply),))
    if not 1 <= max_displacement <= 4:
        raise ValueError("max_displacement must be in 1..4, got %r" % (max_displacement,))


def forward(input1, input2, rbot1, rbot2, output, pad_size, kernel_size, max_displacement, stride1,
            stride2, corr_multiply):
    _check(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
    ops.corr2d_forward_into(input1, input2, output, max_displacement)
    return 1


def backward(input1, input2, rbot1, rbot2, grad_output, grad_input1, grad_input2, pad_size,
             kernel_size, max_displacement, stride1, stride2, corr_multiply):
    _check(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
    ops.corr2d_backward_into(input1, input2, grad_output, grad_input1, grad_input2, max_displacement)
    return 1
