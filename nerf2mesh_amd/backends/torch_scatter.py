"""`torch_scatter.scatter_add` as nerf2mesh uses it (nerf/renderer.py:934-941): 1-D sum-scatter into `out`."""
import torch


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    if out is None:
        n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        shape = list(src.shape)
        shape[dim] = n
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)
