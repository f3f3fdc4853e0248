import torch


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def to_nhwc(x, cpad=None):
    """NCHW cpu tensor -> NHWC (channel-padded) tensor"""
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > x.shape[-1]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
    return x.contiguous()


def to_nchw(x, c=None):
    x = x.permute(0, 3, 1, 2)
    if c is not None:
        x = x[:, :c]
    return x.contiguous()
