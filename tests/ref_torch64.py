"""Independent fp64 re-derivations of the hot-path ops, written from the MATH in SURVEY.md Appendix A
(not from the kernels), used to cross-check the C oracle.  torch.float64 + autograd gives the
gradients the backward kernels must reproduce."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def corr_shape(H, W, pad, K, md, s1, s2):
    kr = (K - 1) // 2
    border = md + kr
    topH = math.ceil((H + 2 * pad - 2 * border) / s1)
    topW = math.ceil((W + 2 * pad - 2 * border) / s1)
    ngr = md // s2
    return (2 * ngr + 1) ** 2, topH, topW, ngr


def correlation(b0: torch.Tensor, b1: torch.Tensor, pad, K, md, s1, s2, subtract=False) -> torch.Tensor:
    """top[n,(q,o),y,x] = 1/(K*K*C) * sum_{j,i,c} P0[n,c,y1+j,x1+i] (*|-) P1[n,c,y1+j+q*s2,x1+i+o*s2]
    with y1 = y*s1 + md, x1 = x*s1 + md in PADDED coordinates (Appendix A.1)."""
    N, C, H, W = b0.shape
    topC, topH, topW, ngr = corr_shape(H, W, pad, K, md, s1, s2)
    P0 = F.pad(b0, (pad, pad, pad, pad))
    P1 = F.pad(b1, (pad, pad, pad, pad))
    outs = []
    for q in range(-ngr, ngr + 1):
        for o in range(-ngr, ngr + 1):
            acc = 0
            for j in range(K):
                for i in range(K):
                    ys = md + j
                    xs = md + i
                    a = P0[:, :, ys: ys + (topH - 1) * s1 + 1: s1, xs: xs + (topW - 1) * s1 + 1: s1]
                    b = P1[:, :, ys + q * s2: ys + q * s2 + (topH - 1) * s1 + 1: s1,
                           xs + o * s2: xs + o * s2 + (topW - 1) * s1 + 1: s1]
                    acc = acc + ((a - b).abs().sum(1) if subtract else (a * b).sum(1))
            outs.append(acc / (K * K * C))
    return torch.stack(outs, 1)


def correlation1d(b0: torch.Tensor, b1: torch.Tensor, pad, K, md, s1, s2, single_direction=0, subtract=False) -> torch.Tensor:
    """Horizontal cost volume: top[n,c,y,x] = 1/(K*K*C) * sum_{j,i,ch} P0[n,ch,y*s1+j,x1+i] (*|-) P1[n,ch,y*s1+j,x1+i+(c+x_shift)*s2],
    x1 = x*s1 + md in x-PADDED coordinates (padding in x only); x_shift = -md//s2 (both), 0 (right), -(md//s2 + 1) (left).
    Positions outside the padded row count as zeros (what the reference's flat indexing reads whenever pad >= the overshoot)."""
    N, C, H, W = b0.shape
    kr = (K - 1) // 2
    ngr = md // s2
    ngw = ngr + 1 if single_direction != 0 else 2 * ngr + 1
    xshift = -ngw if single_direction == -1 else (0 if single_direction == 1 else -ngr)
    topW = math.ceil((W + 2 * pad - 2 * (md + kr)) / s1)
    topH = math.ceil((H - 2 * kr) / s1)
    extra = max(0, -(md + xshift * s2))          # left mode starts one grid step beyond the radius
    P0 = F.pad(b0, (pad + extra, pad, 0, 0))
    P1 = F.pad(b1, (pad + extra, pad, 0, 0))
    outs = []
    for c in range(ngw):
        d = (c + xshift) * s2
        acc = 0
        for j in range(K):
            for i in range(K):
                xs = md + i + extra
                a = P0[:, :, j: j + (topH - 1) * s1 + 1: s1, xs: xs + (topW - 1) * s1 + 1: s1]
                b = P1[:, :, j: j + (topH - 1) * s1 + 1: s1, xs + d: xs + d + (topW - 1) * s1 + 1: s1]
                acc = acc + ((a - b).abs().sum(1) if subtract else (a * b).sum(1))
        outs.append(acc / (K * K * C))
    return torch.stack(outs, 1)


def flow_warp(image: torch.Tensor, flow: torch.Tensor, fill=0.0) -> torch.Tensor:
    """Appendix A.3: bilinear sample at (x+u, y+v), right/bottom neighbour clamped, fill outside."""
    N, C, H, W = image.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=image.dtype), torch.arange(W, dtype=image.dtype), indexing="ij")
    x2 = xs[None] + flow[:, 0]
    y2 = ys[None] + flow[:, 1]
    inb = (x2 >= 0) & (y2 >= 0) & (x2 < W) & (y2 < H)
    x2c = torch.where(inb, x2, torch.zeros_like(x2))
    y2c = torch.where(inb, y2, torch.zeros_like(y2))
    xl = x2c.detach().floor().long()
    yt = y2c.detach().floor().long()
    xr = torch.clamp(xl + 1, max=W - 1)
    yb = torch.clamp(yt + 1, max=H - 1)
    a = (x2c - xl)[:, None]
    b = (y2c - yt)[:, None]
    flat = image.reshape(N, C, H * W)

    def g(yy, xx):
        idx = (yy * W + xx).reshape(N, 1, H * W).expand(N, C, H * W)
        return torch.gather(flat, 2, idx).reshape(N, C, H, W)

    out = (1 - a) * (1 - b) * g(yt, xl) + a * (1 - b) * g(yt, xr) + (1 - a) * b * g(yb, xl) + a * b * g(yb, xr)
    return torch.where(inb[:, None], out, torch.full_like(out, fill))


def _tri(t):
    return np.where((t >= -1) & (t < 0), t + 1, np.where((t >= 0) & (t <= 1), 1 - t, 0.0))


def _cub(t):
    x = np.abs(t)
    return np.where(x <= 1, x * x * (1.5 * x - 2.5) + 1, np.where(x < 2, x * (x * (-0.5 * x + 2.5) - 4) + 2, 0.0))


def resample(x: np.ndarray, Hout, Wout, kind="linear", antialias=True) -> np.ndarray:
    """Appendix A.5 in float64 (including the fx/fy swap of the half-pixel offsets)."""
    N, C, Hin, Win = x.shape
    fx = np.float32(Win) / np.float32(Wout)
    fy = np.float32(Hin) / np.float32(Hout)
    fx, fy = float(fx), float(fy)
    out = np.zeros((N, C, Hout, Wout))
    aa = ((fx > 1) or (fy > 1)) and antialias
    ax = 1.0 / (fx if aa else 1.0)
    ay = 1.0 / (fy if aa else 1.0)
    kw = 4 if kind == "cubic" else 2
    rx = 2 if fx < 1 else math.ceil(kw / ax)
    ry = 2 if fy < 1 else math.ceil(kw / ay)
    k = _cub if kind == "cubic" else _tri
    for yo in range(Hout):
        y_in = yo * fy + fx / 2 - 0.5
        yr = int(math.floor(abs(y_in) + 0.5) * (1 if y_in >= 0 else -1))
        for xo in range(Wout):
            x_in = xo * fx + fy / 2 - 0.5
            xr = int(math.floor(abs(x_in) + 0.5) * (1 if x_in >= 0 else -1))
            if kind == "nearest":
                out[:, :, yo, xo] = x[:, :, min(max(yr, 0), Hin - 1), min(max(xr, 0), Win - 1)]
                continue
            ys = np.arange(max(yr - ry, 0), min(yr + ry, Hin - 1) + 1)
            xs = np.arange(max(xr - rx, 0), min(xr + rx, Win - 1) + 1)
            wy = ay * k(ay * (y_in - ys))
            wx = ax * k(ax * (x_in - xs))
            w = wy[:, None] * wx[None, :]
            ws = w.sum()
            if ws == 0:
                continue
            out[:, :, yo, xo] = (x[:, :, ys[0]: ys[-1] + 1, xs[0]: xs[-1] + 1] * w).sum((2, 3)) / ws
    return out


def l1loss(b0: torch.Tensor, b1, l2_per_location, l2_prescale, normalize_by_num_entries, epsilon, plateau):
    """Appendix A.6.  Returns (loss, normalize_coeff)."""
    d = b0 - b1 if b1 is not None else b0
    N, C = d.shape[:2]
    mask = ~torch.isnan(d)
    norm = mask.sum().to(d.dtype) / C if normalize_by_num_entries else torch.tensor(float(N), dtype=d.dtype)
    d = torch.where(mask, d, torch.zeros_like(d))
    if l2_per_location:
        w = 1.0 / C if l2_prescale else 1.0
        s = (d * d).sum(1) * w
        if plateau > 0:
            s = torch.where(s.detach() < plateau * plateau, torch.zeros_like(s), s)
        e = torch.sqrt(s + epsilon)
        return e.sum() / norm, norm
    if plateau > 0:
        d = torch.where(d.detach().abs() < plateau, torch.zeros_like(d), d)
    return d.abs().sum() / norm, norm


def channel_norm(x: torch.Tensor) -> torch.Tensor:
    return torch.sqrt((x * x).sum(1, keepdim=True))


def downsample(x: np.ndarray, Hout, Wout) -> np.ndarray:
    """Appendix A.8."""
    N, C, Hin, Win = x.shape
    if (Hin, Win) == (Hout, Wout):
        return x.astype(np.float64)
    ws = np.float32(Win - 1) / np.float32(Wout - 1)
    hs = np.float32(Hin - 1) / np.float32(Hout - 1)
    wr, hr = math.ceil(ws), math.ceil(hs)
    out = np.zeros((N, C, Hout, Wout))
    for dy in range(Hout):
        boty = float(np.float32(np.float32(dy) / np.float32(Hout - 1)) * np.float32(Hin - 1))
        iy = int(math.floor(boty + 0.5))
        for dx in range(Wout):
            botx = float(np.float32(np.float32(dx) / np.float32(Wout - 1)) * np.float32(Win - 1))
            ix = int(math.floor(botx + 0.5))
            val = np.zeros((N, C))
            wsum = np.zeros((N, C))
            nansum = np.zeros((N, C))
            for by in range(iy - hr, iy + hr + 1):
                for bx in range(ix - wr, ix + wr + 1):
                    if 0 <= bx < Win and 0 <= by < Hin:
                        s = x[:, :, by, bx].astype(np.float64)
                        w = max(0.0, 1 - abs(bx - botx) / float(ws)) * max(0.0, 1 - abs(by - boty) / float(hs))
                        isn = np.isnan(s)
                        nansum += np.where(isn, w, 0.0)
                        val += np.where(isn, 0.0, s * w)
                        wsum += np.where(isn, 0.0, w)
            with np.errstate(divide="ignore", invalid="ignore"):
                r = val / wsum
                r[(nansum / wsum) > 0.5] = np.nan
            out[:, :, dy, dx] = r
    return out
