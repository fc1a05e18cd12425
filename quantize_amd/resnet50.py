"""The ResNet-50 conv stack the headline metric is quoted on (BASELINE.json).

53 convolutions of torchvision's ResNet-50 v1.5 (stride on the 3x3), as a flat list of
independent conv problems: each layer gets its own synthetic input of the right shape
(there is no ReLU/BN/requantisation in the reference's op contract, so the stack is not a
trained network).  Synthetic operand distributions follow SURVEY.md section 8d.
"""
from collections import namedtuple

ConvLayer = namedtuple("ConvLayer", "name IC OC K stride pad H")  # square input H x H


def conv_layers():
    layers = [ConvLayer("conv1", 3, 64, 7, 2, 3, 224)]
    inplanes, H = 64, 56  # after the stem's maxpool
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            pre = "layer%d.%d." % (li, b)
            layers.append(ConvLayer(pre + "conv1", inplanes, planes, 1, 1, 0, H))
            layers.append(ConvLayer(pre + "conv2", planes, planes, 3, s, 1, H))
            Ho = (H + 2 - 3) // s + 1
            layers.append(ConvLayer(pre + "conv3", planes, planes * 4, 1, 1, 0, Ho))
            if b == 0:
                layers.append(ConvLayer(pre + "downsample", inplanes, planes * 4, 1, s, 0, H))
            inplanes, H = planes * 4, Ho
    return layers


def out_size(layer):
    return (layer.H + 2 * layer.pad - layer.K) // layer.stride + 1


def macs_per_image(layer):
    o = out_size(layer)
    return layer.OC * o * o * layer.IC * layer.K * layer.K


def algorithmic_bytes(layer, N, x_bits=8, w_bits=8, float_input=False):
    """Bytes the op's contract moves once: activations in + weights in + fp32 output out."""
    o = out_size(layer)
    n_in = N * layer.IC * layer.H * layer.H
    n_w = layer.OC * layer.IC * layer.K * layer.K
    in_b = n_in * 4 if float_input else (n_in * x_bits + 7) // 8
    return in_b + (n_w * w_bits + 7) // 8 + N * layer.OC * o * o * 4


def synth_layer(layer, idx, N, device, x_bits=8, w_bits=8, asymmetric=False, with_bias=True, rank=0):
    """Synthetic operands of one layer: seed 1000+idx; q_w ~ U{qmin..qmax},
    s_w[oc] ~ U(2.5e-4, 7.5e-4), z_w = 0; q_x ~ U{qmin..qmax}, per-tensor s_x = 2e-3, z_x = 0
    (asymmetric: unsigned q_x with z_x = 133.2578 * 2^(bits-8), kernel convention); bias ~ N(0, 0.1).
    Weights, scales and bias are identical on every rank (replicated); the activations of rank r > 0
    (its own 256-image shard of the global batch) come from a rank-specific seed.
    Returns integer-valued int16 tensors (to be packed by the engine) plus fp32 parameters."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(1000 + idx)
    wlo, whi = -(1 << (w_bits - 1)), (1 << (w_bits - 1)) - 1
    qw = torch.randint(wlo, whi + 1, (layer.OC, layer.IC, layer.K, layer.K), generator=g, device=device,
                       dtype=torch.int16)
    sw = (torch.rand(layer.OC, generator=g, device=device) * 5e-4 + 2.5e-4).reshape(-1, 1, 1, 1)
    zw = torch.zeros_like(sw)
    bias = (torch.randn(layer.OC, generator=g, device=device) * 0.1) if with_bias else None
    if rank:
        g.manual_seed(1000 + idx + 1000003 * rank)
    if asymmetric:
        qx = torch.randint(0, 1 << x_bits, (N, layer.IC, layer.H, layer.H), generator=g, device=device,
                           dtype=torch.int16)
        zx = torch.full((1,), 133.2578 * (2.0 ** (x_bits - 8)), device=device)
        x_sign = False
    else:
        xlo, xhi = -(1 << (x_bits - 1)), (1 << (x_bits - 1)) - 1
        qx = torch.randint(xlo, xhi + 1, (N, layer.IC, layer.H, layer.H), generator=g, device=device,
                           dtype=torch.int16)
        zx = torch.zeros(1, device=device)
        x_sign = True
    sx = torch.full((1,), 2e-3, device=device)
    return dict(qx=qx, x_sign=x_sign, sx=sx, zx=zx, qw=qw, w_sign=True, sw=sw, zw=zw, bias=bias)
