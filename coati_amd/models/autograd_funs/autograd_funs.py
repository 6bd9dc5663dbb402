"""Differentiable all-gather with the reference's interface (coati/models/autograd_funs/autograd_funs.py:5-25):
forward = all_gather + rank-major cat, backward = reduce_scatter(sum) of the fp32 gradient chunks.
Pure torch.distributed (RCCL on the GPUs, gloo in the CPU tests); one collective each way instead of the
reference's per-rank tensor list."""
import torch
import torch.distributed as dist


class AllGatherFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor: torch.Tensor, reduce_dtype: torch.dtype = torch.float32):
        ctx.reduce_dtype = reduce_dtype
        W = dist.get_world_size()
        out = torch.empty((W * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(out, tensor.contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        W = dist.get_world_size()
        g = grad_output.to(ctx.reduce_dtype).contiguous()
        out = torch.empty((g.shape[0] // W,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out.to(grad_output.dtype), None


def all_gather(tensor):
    return AllGatherFunction.apply(tensor)
