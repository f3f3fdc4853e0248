"""Thin torch-tensor wrappers over the single-operator entry points of libd3f_hip.so.

Activations are NHWC torch tensors ([B, H, W, C], C padded to 4 for f32 / 8 for bf16) that live
on the HIP device; weights are the f32 torch-layout masters.  These wrappers exist for the
parity tests and for callers that want one kernel at a time; the training path goes through
`Unet` (one C call per forward / backward).  No fallbacks: every function launches a HIP kernel.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, stream_ptr

F32, BF16, F32X3 = _lib.F32, _lib.BF16, _lib.F32X3


def _tdtype(dtype):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def _dev(t):
    if t.device.type != "cuda":
        raise _lib.D3FError("libd3f_hip ops need tensors on the HIP device (no CPU fallback)")
    return t.device


def make_desc(B, H, W, C0, C1, Cout, k, stride, pad, upsample0=False, cin_real=None):
    return ConvDesc(B, H, W, C0, C1, 1 if upsample0 else 0, Cout, k, k, stride, pad,
                    cin_real if cin_real is not None else C0 + C1)


def out_hw(d):
    return (d.H + 2 * d.pad - d.KH) // d.stride + 1, (d.W + 2 * d.pad - d.KW) // d.stride + 1


def nchw_to_nhwc(x, cpad, dtype=F32):
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, cpad), dtype=_tdtype(dtype), device=_dev(x))
    check(_lib.lib().d3f_nchw_to_nhwc(dtype, ptr(x.contiguous().float()), ptr(out), B, Cc, H, W, cpad, stream_ptr()))
    return out


def nhwc_to_nchw(x, c, dtype=F32):
    B, H, W, cpad = x.shape
    out = torch.empty((B, c, H, W), dtype=torch.float32, device=_dev(x))
    check(_lib.lib().d3f_nhwc_to_nchw(dtype, ptr(x.contiguous()), ptr(out), B, c, H, W, cpad, stream_ptr()))
    return out


def pack_weights(d, w, dtype=F32, dgrad=True):
    L = _lib.lib()
    dev = _dev(w)
    wf = torch.empty(L.d3f_conv_packed_bytes(dtype, C.byref(d), 0), dtype=torch.uint8, device=dev)
    wd = torch.empty(L.d3f_conv_packed_bytes(dtype, C.byref(d), 1), dtype=torch.uint8, device=dev) if dgrad else None
    check(L.d3f_conv_pack_weights(dtype, C.byref(d), ptr(w.contiguous().float()), ptr(wf), ptr(wd), stream_ptr()))
    return wf, wd


def _conv_ws(d, dtype, which, dev, splitk):
    if not splitk:
        return None
    n = _lib.lib().d3f_conv_workspace_bytes(dtype, C.byref(d), which)
    return torch.empty(max(n, 16), dtype=torch.uint8, device=dev)


def conv_forward(d, src0, src1, wf, dtype=F32, want_stats=True, splitk=False):
    """splitk=True hands the kernel a workspace so small-M layers may split their K loop."""
    L = _lib.lib()
    ho, wo = out_hw(d)
    y = torch.empty((d.B, ho, wo, d.Cout), dtype=_tdtype(dtype), device=_dev(src0))
    ws = _conv_ws(d, dtype, 0, y.device, splitk)
    tiles = C.c_int()
    n = L.d3f_conv_stats_floats(dtype, C.byref(d), int(splitk), C.byref(tiles))
    stats = torch.zeros(n, dtype=torch.float32, device=y.device) if want_stats else None
    check(L.d3f_conv_forward(dtype, C.byref(d), ptr(src0), ptr(src1), ptr(wf), ptr(y), ptr(stats), ptr(ws),
                             stream_ptr()))
    return y, stats, tiles.value


def conv_winograd_applies(d, dtype=F32):
    """does the whole-network plan run this layer's forward as Winograd F(2x2, 3x3) (conv_winograd.hip)?"""
    return bool(_lib.lib().d3f_conv_winograd_applies(dtype, C.byref(d)))


def conv_winograd_pack(d, w):
    L = _lib.lib()
    n = L.d3f_conv_winograd_filter_bytes(C.byref(d))
    if n == 0:
        raise _lib.D3FError("conv_winograd_pack: the layer does not fit the Winograd kernel")
    u = torch.empty(n, dtype=torch.uint8, device=_dev(w))
    check(L.d3f_conv_winograd_pack(C.byref(d), ptr(w.contiguous().float()), ptr(u), stream_ptr()))
    return u


def conv_winograd_forward(d, src0, u, want_stats=True, scale=None, shift=None, residual=None, relu=False):
    """fp32 Winograd forward on its own.  scale / shift given: the eval epilogue relu?(y * scale + shift + residual?);
    else the raw conv output and one (sum, sumsq) statistics row per workgroup -> (y, stats, tiles)."""
    L = _lib.lib()
    y = torch.empty((d.B, d.H, d.W, d.Cout), dtype=torch.float32, device=_dev(src0))
    tiles = C.c_int()
    n = L.d3f_conv_winograd_stats_floats(C.byref(d), C.byref(tiles))
    stats = torch.zeros(n, dtype=torch.float32, device=y.device) if (want_stats and scale is None) else None
    check(L.d3f_conv_winograd_forward(C.byref(d), ptr(src0), ptr(u), ptr(y), ptr(stats), ptr(scale), ptr(shift),
                                      ptr(residual), int(relu), stream_ptr()))
    return y, stats, tiles.value


def conv_upsample_folded(d, dtype=F32):
    """does this conv(cat(upsample2x(src0), src1)) run with the up-sampling folded into pre-summed weights?"""
    return bool(d.upsample0) and bool(_lib.lib().d3f_conv_upsample_folded(dtype, C.byref(d)))


def conv_backward_data(d, dy, wd, dtype=F32, dx0=None, dx1=None, acc0=False, acc1=False, splitk=False, summed=None):
    """(dx0, dx1): gradients of src0 and src1.  For an up-sampled src0 (d.upsample0) dx0 is the gradient of the
    LOW-resolution tensor [B, H/2, W/2, C0] -- written directly by the folded 4x4 stride-2 kernel where the layer
    qualifies, else reduced here from the full-resolution gradient of the up-sampled operand.  summed=False keeps the
    full-resolution + d3f_upsample2x_backward route also where the launch could sum the 2x2 blocks itself."""
    dev = _dev(dy)
    ws = _conv_ws(d, dtype, 1, dev, splitk)
    folded = bool(d.upsample0) and bool(_lib.lib().d3f_conv_upsample_folded(dtype, C.byref(d)))
    # ... or summed 2x2 in the launch's own epilogue (the patch form of the bf16 16 -> 32 layer): low resolution too.
    # That form is opt-in (upsample0 = 2 in the descriptor): a C caller on the full-resolution contract keeps it.
    if bool(d.upsample0) and not folded and summed is not False and \
            bool(_lib.lib().d3f_conv_upsample_summed(dtype, C.byref(d))):
        d = ConvDesc(*[getattr(d, n) for n, _ in ConvDesc._fields_])
        d.upsample0 = 2
        folded = True
    low = (d.B, d.H // 2, d.W // 2, d.C0)
    if d.upsample0 and not folded:
        if acc0 or dx0 is not None:
            raise ValueError("conv_backward_data: an up-sampled source that is not folded takes no dx0 / acc0")
        full = torch.empty((d.B, d.H, d.W, d.C0), dtype=_tdtype(dtype), device=dev)
    elif dx0 is None:
        full = None
        dx0 = torch.empty(low if d.upsample0 else (d.B, d.H, d.W, d.C0), dtype=_tdtype(dtype), device=dev)
    else:
        full = None
    if dx1 is None and d.C1 > 0:
        dx1 = torch.empty((d.B, d.H, d.W, d.C1), dtype=_tdtype(dtype), device=dev)
    check(_lib.lib().d3f_conv_backward_data(dtype, C.byref(d), ptr(dy), ptr(wd), ptr(full if full is not None else dx0),
                                            ptr(dx1), int(acc0), int(acc1), ptr(ws), stream_ptr()))
    if full is not None:
        dx0 = upsample2x_backward(full, dtype)
    return dx0, dx1


def conv_backward_weight(d, dy, src0, src1, dtype=F32):
    L = _lib.lib()
    ws = torch.empty(L.d3f_conv_backward_weight_workspace_bytes(dtype, C.byref(d)), dtype=torch.uint8, device=_dev(dy))
    dw = torch.empty((d.Cout, d.CinReal, d.KH, d.KW), dtype=torch.float32, device=dy.device)
    check(L.d3f_conv_backward_weight(dtype, C.byref(d), ptr(dy), ptr(src0), ptr(src1), ptr(ws), ptr(dw), stream_ptr()))
    return dw


def bn_finalize(stats, tiles, Cc, count, gamma, beta, running_mean=None, running_var=None):
    coef = torch.empty(4 * Cc, dtype=torch.float32, device=_dev(stats))
    check(_lib.lib().d3f_bn_finalize(ptr(stats), tiles, Cc, count, ptr(gamma), ptr(beta), ptr(running_mean),
                                     ptr(running_var), ptr(coef), stream_ptr()))
    return coef


def bn_apply(y, coef, residual=None, relu=True, dtype=F32):
    Cc = y.shape[-1]
    out = torch.empty_like(y)
    check(_lib.lib().d3f_bn_apply(dtype, ptr(y), ptr(coef), Cc, y.numel() // Cc, ptr(residual), int(relu),
                                  ptr(out), stream_ptr()))
    return out


def bn_backward(dA, a, y, coef, gamma, want_dres=False, dtype=F32):
    L = _lib.lib()
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    ws = torch.empty(L.d3f_bn_backward_workspace_bytes(dtype, Cc, rows), dtype=torch.uint8, device=_dev(y))
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    dgamma = torch.empty(Cc, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=y.device)
    check(L.d3f_bn_backward(dtype, ptr(dA), ptr(a), ptr(y), ptr(coef), ptr(gamma), Cc, rows, ptr(dy), ptr(dres),
                            ptr(dgamma), ptr(dbeta), ptr(ws), stream_ptr()))
    return dy, dres, dgamma, dbeta


def maxpool_forward(x, dtype=F32):
    B, H, W, Cc = x.shape
    out = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=_dev(x))
    idx = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.uint8, device=x.device)
    check(_lib.lib().d3f_maxpool3x3s2_forward(dtype, ptr(x), ptr(out), ptr(idx), B, H, W, Cc, stream_ptr()))
    return out, idx


def maxpool_backward(dout, idx, H, W, dtype=F32, din=None):
    B, _, _, Cc = dout.shape
    acc = din is not None
    if din is None:
        din = torch.empty((B, H, W, Cc), dtype=dout.dtype, device=_dev(dout))
    check(_lib.lib().d3f_maxpool3x3s2_backward(dtype, ptr(dout), ptr(idx), ptr(din), int(acc), B, H, W, Cc, stream_ptr()))
    return din


def upsample2x_backward(dfull, dtype=F32):
    B, H, W, Cc = dfull.shape
    out = torch.empty((B, H // 2, W // 2, Cc), dtype=dfull.dtype, device=_dev(dfull))
    check(_lib.lib().d3f_upsample2x_backward(dtype, ptr(dfull), ptr(out), B, H // 2, W // 2, Cc, stream_ptr()))
    return out


def affine_warp(x, theta):
    """affine_grid + grid_sample(bilinear, zeros, align_corners=False) of NCHW f32 images, theta [B, 2, 3]."""
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    theta = theta.to(device=_dev(x), dtype=torch.float32).contiguous()
    if theta.shape != (B, 2, 3):
        raise ValueError(f"theta must be [{B}, 2, 3], got {list(theta.shape)}")
    out = torch.empty_like(x)
    check(_lib.lib().d3f_affine_warp(ptr(x), ptr(theta), ptr(out), B, Cc, H, W, stream_ptr()))
    return out


def u8rgb_normalise(frames, mean, std):
    """uint8 RGB [B, H, W, 3] on the HIP device -> normalised NCHW float32 ((u8 / 255 - mean) / std per channel): the
    host transform NormalizeToTensor bit for bit (d3f_u8rgb_normalise)"""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError("u8rgb_normalise expects uint8 frames [B, H, W, 3] (RGB)")
    frames = frames.contiguous()
    B, H, W, _ = frames.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=_dev(frames))
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().d3f_u8rgb_normalise(ptr(frames), ptr(out), B, H, W, m, s, stream_ptr()))
    return out


def noise_blend(x, noise, y_uniform, lam, return_r=False):
    x = x.contiguous().float()
    out = torch.empty_like(x)
    B = x.shape[0]
    r = torch.empty(B, dtype=torch.float32, device=_dev(x)) if return_r else None
    check(_lib.lib().d3f_noise_blend(ptr(x), ptr(noise.contiguous().float()), ptr(y_uniform.contiguous().float()),
                                     float(lam), ptr(out), ptr(r), B, x.numel() // max(B, 1), stream_ptr()))
    return (out, r) if return_r else out


def noise_blend_fixed(x, noise, ratio):
    """sqrt(1-r)*x + sqrt(r)*noise with one ratio for every image (float) or a per-image tensor [B]."""
    x = x.contiguous().float()
    B = x.shape[0]
    r = ratio if torch.is_tensor(ratio) else torch.ones(B, device=_dev(x)) * float(ratio)
    r = r.to(device=_dev(x), dtype=torch.float32).reshape(-1).contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().d3f_noise_blend_fixed(ptr(x), ptr(noise.contiguous().float()), ptr(r), ptr(out), B,
                                           x.numel() // max(B, 1), stream_ptr()))
    return out


def l1_per_image(pred, target):
    """mean |pred - target| over each image: [B] f32."""
    L = _lib.lib()
    pred, target = pred.contiguous().float(), target.contiguous().float()
    B = pred.shape[0]
    ws = torch.empty(L.d3f_l1_per_image_workspace_bytes(B), dtype=torch.uint8, device=_dev(pred))
    out = torch.empty(B, dtype=torch.float32, device=pred.device)
    check(L.d3f_l1_per_image(ptr(pred), ptr(target), ptr(out), ptr(ws), B, pred.numel() // max(B, 1), stream_ptr()))
    return out


def mse_ssim_loss(pred, target, in_min=-1.0, in_max=1.0):
    """returns (loss[3] = {loss, mse, ssim} device tensor, grad wrt pred)."""
    L = _lib.lib()
    B, Cc, H, W = pred.shape
    if Cc != 3:
        raise ValueError("SSIM is defined for 3-channel images (piqa n_channels=3)")
    pred = pred.contiguous().float()
    target = target.contiguous().float()
    ws = torch.empty(L.d3f_mse_ssim_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=_dev(pred))
    out = torch.empty(3, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred)
    check(L.d3f_mse_ssim_loss(ptr(pred), ptr(target), float(in_min), float(in_max), ptr(out), ptr(grad), ptr(ws),
                              B, H, W, stream_ptr()))
    return out, grad


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    check(_lib.lib().d3f_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step,
                                   grad_scale, stream_ptr()))


def ema_lerp(ema, online, weight):
    check(_lib.lib().d3f_ema_lerp(ptr(ema), ptr(online), ema.numel(), float(weight), stream_ptr()))
