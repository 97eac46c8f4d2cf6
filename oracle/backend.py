"""CPU 'backend' with the call signature of flownet2_amd.functional, implemented by the C oracle
(forward only).  TEST INFRASTRUCTURE: lets tests / smoke / bench's cpu_baseline run the same graph
(flownet2_amd.nets) on the host with torch-CPU convolutions (the reference's CPU conv is
im2col + cblas_sgemm, base_conv_layer.cpp:255-272 -- mathematically the same contraction)."""
from __future__ import annotations

import numpy as np
import torch

import oracle


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False)


def correlation(b0, b1, pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1, correlation_type=0):
    p = oracle.corr_params(pad, kernel_size, max_displacement, stride_1, stride_2, correlation_type)
    return torch.from_numpy(oracle.correlation_forward(p, _np(b0), _np(b1)))


def flow_warp(image, flow, fill_value=oracle.FILL_ZERO):
    return torch.from_numpy(oracle.flow_warp_forward(_np(image), _np(flow), fill_value))


def resample(x, height, width, type=oracle.LINEAR, antialias=True):
    return torch.from_numpy(oracle.resample_forward(_np(x), height, width, type, antialias))


def downsample(x, top_height, top_width):
    return torch.from_numpy(oracle.downsample_forward(_np(x), top_height, top_width))


def channel_norm(x):
    return torch.from_numpy(oracle.channel_norm_forward(_np(x)))


def l1_loss(b0, b1=None, l2_per_location=False, l2_prescale_by_channels=False, normalize_by_num_entries=False,
            epsilon=1e-2, plateau=0.0):
    p = oracle.l1_params(l2_per_location, l2_prescale_by_channels, normalize_by_num_entries, epsilon, plateau)
    loss, _ = oracle.l1loss_forward(p, _np(b0), _np(b1) if b1 is not None else None)
    return torch.tensor(loss)


def predict_flow_conv(x, weight, bias=None):
    return torch.from_numpy(oracle.predict_flow_conv_forward(_np(x), _np(weight), _np(bias) if bias is not None else None))


def upsample_flow_deconv(x, weight, bias=None):
    return torch.from_numpy(oracle.upsample_flow_deconv_forward(_np(x), _np(weight), _np(bias) if bias is not None else None))


def conv_bias_leaky_relu(y, bias, negative_slope=0.1):
    return torch.from_numpy(oracle.bias_leaky_relu_forward(_np(y), _np(bias) if bias is not None else None, negative_slope))
