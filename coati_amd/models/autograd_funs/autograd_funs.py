"""Differentiable all-gather behind the reference's import path and names (coati/models/autograd_funs/
autograd_funs.py:5-25).  The collectives themselves live in coati_amd.distributed (one all_gather_into_tensor forward,
one reduce_scatter_tensor backward -- RCCL on the GPUs, gloo in the CPU tests); this module only adapts them to
torch.autograd so that code written against the reference (`from coati.models.autograd_funs.autograd_funs import
all_gather`) keeps working."""
import torch

from ...distributed import all_gather_cat, reduce_scatter_sum


class AllGatherFunction(torch.autograd.Function):
    """forward: rows of every rank, rank-major; backward: this rank's row block of the summed gradient (fp32 on the
    wire unless `reduce_dtype` says otherwise, cast back to the incoming gradient's dtype)."""

    @staticmethod
    def forward(ctx, tensor: torch.Tensor, reduce_dtype: torch.dtype = torch.float32):
        ctx.wire_dtype = reduce_dtype
        return all_gather_cat(tensor)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        local = reduce_scatter_sum(grad_output.to(ctx.wire_dtype))
        return local.to(grad_output.dtype), None


def all_gather(tensor):
    return AllGatherFunction.apply(tensor)
