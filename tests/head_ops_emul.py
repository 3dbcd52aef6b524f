"""Torch-CPU emulation of the DPT-head entries of omnivggt-official_amd/ops.py (test infrastructure).

Same signatures and NHWC semantics as ops.head_layernorm / conv / upsample / dpt_out, computed with
ATen ops in the dtype of the inputs. Two uses:
  * CPU tests: run heads_hip.HipDPTHead on this emulation and compare with the PyTorch DPTHead
    (= oracle-checked restatement of heads/dpt_head.py) -> validates weight packing (tap order,
    ConvTranspose scatter), ReLU folding, the out_conv / upsample commutation and the separable UV tables;
  * GPU tests: per-kernel reference for the HIP entries.
"""
import torch
import torch.nn.functional as F


def head_layernorm(x, weight, bias, eps, dtype, views, tokens_per_view=1374, n_special=5):
    x = x.reshape(views, tokens_per_view, -1)[:, n_special:].reshape(views * (tokens_per_view - n_special), -1)
    return F.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).to(dtype)


def _add_pos(y, pos):
    half = y.shape[-1] // 2
    y[..., :half] += pos[0].float()[None, None, :, :]
    y[..., half:] += pos[1].float()[None, :, None, :]
    return y


def conv(x, w, bias, dtype, cout, ksize=1, stride=1, upshuffle=0, relu=False, add1=None, add2=None, pos=None, out_f32=False):
    n, H, W, cin = x.shape
    xf, wf = x.float(), w.float()
    if upshuffle > 1:
        s = upshuffle
        y = (xf.reshape(-1, cin) @ wf.t()).reshape(n, H, W, s, s, cout).permute(0, 1, 3, 2, 4, 5).reshape(n, H * s, W * s, cout)
        if bias is not None:
            y = y + bias.float()
    else:
        wt = wf[:cout].reshape(cout, ksize, ksize, cin).permute(0, 3, 1, 2)
        y = F.conv2d(xf.permute(0, 3, 1, 2), wt, None if bias is None else bias.float(), stride=stride, padding=ksize // 2)
        y = y.permute(0, 2, 3, 1).contiguous()
    if pos is not None:
        y = _add_pos(y, pos)
    if add1 is not None:
        y = y + add1.float()
    if add2 is not None:
        y = y + add2.float()
    if relu:
        y = F.relu(y)
    return y if out_f32 else y.to(dtype)


def upsample(x, OH, OW, dtype, pos=None):
    y = F.interpolate(x.float().permute(0, 3, 1, 2), size=(OH, OW), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    if pos is not None:
        y = _add_pos(y, pos)
    return y.to(dtype)


def dpt_out(h, w2, b2, activation):
    o = h.float() @ w2.float().t() + b2.float()
    val, conf = o[..., :-1], o[..., -1]
    val = torch.exp(val) if activation == "exp" else torch.sign(val) * torch.expm1(val.abs())
    return val.contiguous(), (1 + conf.exp()).contiguous()
