"""Opt-in wiring of packed Quant modules to the operator path.

The reference's packed forward still dequantises and calls F.conv2d / F.linear; the operator call is commented out
(modelzoo/modules/quantconv2d.py:198-210, quantlinear.py:150-161) and the runners' packing loop with it
(runner/ptq.py:106-114).  These two classes are what those call sites become on this engine.  They take the STATE of a
packed module -- tensors only, exactly the state_dict entries pack() leaves behind (quantconv2d.py:187-192:
weight = packed uint8 stream, w_des, w_scale, w_zero, bias, a_quantizer.{scale, zero, qmin, qmax}) -- so no reference
Python is needed where they run, and they keep the weights PACKED (no tunpack on load, the TODO of quantconv2d.py:230-233).

Sign conventions, reconciled here and nowhere else:
  * the modules dequantise as (q + zero) * scale (quantizer.py:218);
  * quantconv2d / quantconv2d_float_input / quantlinear_float_input take (q - zero) * scale -> zeros are negated;
  * quantlinear (packed x packed) takes (q + zero) (quantlinear.cu:115,120)             -> zeros pass unchanged.

Two routes per layer, as the operator's dtype dispatch offers them (quantconv2dop.py:88-95):
  route="packed"  activations are quantised AND packed on the device in one pass (engine.quantize_pack: the
                  Quantizer's round(x / scale - zero).clamp(qmin, qmax) + tpack) and meet the packed weights on the
                  int8 MFMA kernels;
  route="float"   the form the reference's TODO names first ("only support float input and packed weight"): the
                  fake-quantised fp32 activations (q + zero) * scale with the packed weights.
"""
import torch

from . import engine
from .operator import quantconv2d_forward, quantlinear_forward


def _first(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


class _PackedBase:
    def __init__(self, weight, w_des, w_scale, w_zero, bias, a_scale, a_zero, a_qmin, a_qmax, a_bits, a_signed):
        assert weight.dtype == torch.uint8 and weight.dim() == 1, "weight must be the packed 1-D uint8 stream pack() stores"
        self.weight, self.w_des, self.bias = weight, w_des, bias
        self.w_scale, self.w_zero = w_scale.contiguous(), w_zero.contiguous()            # module convention (q + zero)
        self.a_scale = a_scale.reshape(-1).contiguous().float()
        self.a_zero = a_zero.reshape(-1).contiguous().float()
        self.a_qmin, self.a_qmax = float(a_qmin), float(a_qmax)
        self.a_bits, self.a_signed = int(a_bits), bool(a_signed)
        self._neg_w_zero = (-self.w_zero).contiguous()       # kernel convention, built once (tensors stay alive: cache keys)
        self._neg_a_zero = (-self.a_zero).contiguous()

    @classmethod
    def _state(cls, sd, prefix):
        g = lambda k: sd[prefix + k]
        qmin, qmax = float(g("a_quantizer.qmin")), float(g("a_quantizer.qmax"))
        n_levels = int(round(qmax - qmin)) + 1
        a_bits = max(1, (n_levels - 1).bit_length())
        return dict(weight=g("weight"), w_des=g("w_des"), w_scale=g("w_scale"), w_zero=g("w_zero"),
                    bias=sd.get(prefix + "bias"), a_scale=g("a_quantizer.scale"), a_zero=g("a_quantizer.zero"),
                    a_qmin=qmin, a_qmax=qmax, a_bits=a_bits, a_signed=qmin < 0)

    def to(self, device):
        for k, v in list(vars(self).items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def quantize(self, x, channel_dim=1):
        """Quantizer.simulate in packed mode (quantizer.py:215,226) fused with tpack: (packed, des)."""
        return engine.quantize_pack(x.contiguous(), self.a_scale, self.a_zero, self.a_qmin, self.a_qmax, self.a_bits,
                                    self.a_signed, channel_dim)

    def fake_quant(self, x, channel_dim=1):
        """(q + zero) * scale in fp32: what the reference's packed forward feeds F.conv2d (quantconv2d.py:207-208)."""
        shape = [1] * x.dim()
        if self.a_scale.numel() > 1:
            shape[channel_dim] = -1
        s, z = self.a_scale.view(shape), self.a_zero.view(shape)
        q = (x / s - z).round().clamp(self.a_qmin, self.a_qmax)
        return ((q + z) * s).contiguous()


class PackedConv2d(_PackedBase):
    """A packed QuantConv2d's forward on the engine: quantconv2d_forward(x, (weight, w_des, w_scale, w_zero), bias,
    stride, padding, dilation, groups) -- the commented call of quantconv2d.py:204-206 -- with the activation operand
    produced on the device."""

    def __init__(self, *, stride=1, padding=0, **state):
        super().__init__(**state)
        self.stride, self.padding = _first(stride), _first(padding)

    @classmethod
    def from_state_dict(cls, state_dict, prefix="", stride=1, padding=0):
        return cls(stride=stride, padding=padding, **cls._state(state_dict, prefix))

    def __call__(self, x, route="packed"):
        w = (self.weight, self.w_des, self.w_scale, self._neg_w_zero)
        if route == "packed":
            xq, x_des = self.quantize(x)
            return quantconv2d_forward((xq, x_des, self.a_scale, self._neg_a_zero), w, self.bias,
                                       (self.stride, self.stride), (self.padding, self.padding), (1, 1), 1)
        if route == "float":
            return quantconv2d_forward(self.fake_quant(x), w, self.bias, self.stride, self.padding, 1, 1)
        raise ValueError("route must be 'packed' or 'float'")


    # ---- packed in, packed out: the conv-epilogue form of f-2 ----
    def call_packed(self, xq, x_des, consumer=None):
        """Activations as (packed stream, des) -- what quantize() or an earlier call_packed() returned.
        consumer=None: the fp32 output.  consumer = the Packed* layer that reads this layer's output directly: returns the
        (packed, des) pair of ITS activation quantiser, written by this layer's conv kernel itself
        (qe_quantconv2d_requant_prepared): no fp32 tensor between the two layers.  Bit-identical to
        consumer.quantize(self.call_packed(xq, x_des)).  (A ReLU in between folds into the clamp when the consumer's codes
        are unsigned with zero point 0: round(max(y, 0) / s).clamp(0, qmax) == round(y / s).clamp(0, qmax).)"""
        from . import capi
        d = [int(v) for v in x_des.tolist()]
        n_bits, sign, (N, IC, H, W) = d[0], d[1], d[2:6]
        wd = [int(v) for v in self.w_des.tolist()]
        sh = capi.conv_shape(N, IC, H, W, wd[2], wd[4], wd[5], self.stride, self.padding)
        x = capi.qparam(xq, n_bits, sign, self.a_scale, self._neg_a_zero)
        w = capi.qparam(self.weight, wd[0], wd[1], self.w_scale.reshape(-1), self._neg_w_zero.reshape(-1))
        key = (n_bits, capi.conv_prepared_layout(sh, n_bits, wd[0]))
        if getattr(self, "_prep_key", None) != key:      # weights prepared once per table layout (batch size does not enter)
            self._prepared, self._prep_key = capi.conv_prepare(w, self.bias, sh, n_bits), key
        if consumer is None:
            return capi.quantconv2d_prepared(x, w, self.bias, sh, self._prepared)
        rq = capi.requant(consumer.a_scale, consumer.a_zero, consumer.a_qmin, consumer.a_qmax, consumer.a_bits, consumer.a_signed)
        out, status = capi.quantconv2d_requant_prepared(x, w, self.bias, sh, self._prepared, rq)
        if int(status.item()) != 0:
            raise RuntimeError("The input tensor is out of range.")     # tpack.cu:14, as engine.tpack raises it
        OH, OW = capi.out_hw(sh)
        des = torch.tensor([consumer.a_bits, 1 if consumer.a_signed else 0, N, sh.OC, OH, OW], dtype=torch.int32, device=xq.device)
        return out, des


class PackedLinear(_PackedBase):
    """A packed QuantLinear's forward on the engine (quantlinear.py:150-161).  The packed x packed kernel of the
    reference indexes the activation scale by batch ROW (quantlinear.cu:96), so a per-tensor scale is what the modules'
    'layer' granularity provides; inputs with leading dimensions are flattened to (rows, in_features)."""

    @classmethod
    def from_state_dict(cls, state_dict, prefix=""):
        return cls(**cls._state(state_dict, prefix))

    def __call__(self, x, route="packed"):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        ws, wz = self.w_scale.reshape(-1), self.w_zero.reshape(-1)
        if route == "packed":
            assert self.a_scale.numel() == 1, "quantlinear takes a per-tensor or per-row activation scale"
            xq, x_des = self.quantize(x2, channel_dim=x2.dim() - 1)
            y = quantlinear_forward((xq, x_des, self.a_scale, self.a_zero), (self.weight, self.w_des, ws, wz), self.bias)
        elif route == "float":
            y = quantlinear_forward(self.fake_quant(x2, channel_dim=x2.dim() - 1),
                                    (self.weight, self.w_des, ws, self._neg_w_zero.reshape(-1)), self.bias)
        else:
            raise ValueError("route must be 'packed' or 'float'")
        return y.reshape(*lead, y.shape[-1])


class PackedMultiheadAttention:
    """A packed QuantMultiheadAttention's forward on the engine (modelzoo/modules/quantmultiheadattention.py:262-...).
    pack() leaves the q / k / v projection weights and out_proj.weight as packed streams with {q,k,v,out}_proj_{scale,
    zero,des} beside them and one activation quantiser per input (:165-223); the reference's packed forward dequantises
    all four and calls F.multi_head_attention_forward.  Here the three input projections run as packed linears (either
    route of PackedLinear, each with its own activation quantiser and its slice of in_proj_bias), the attention core
    softmax(Q K^T / sqrt(d)) V stays fp32 torch as in the reference, and out_proj -- whose input has no quantiser in the
    reference -- takes the fp32 x packed-weight operator (quantlinear_float_input).  Inputs are (L, N, E) / (S, N, kdim)
    (batch_first=False, the module's default); returns (attn_output, averaged attention weights or None)."""

    def __init__(self, q, k, v, out_weight, out_des, out_scale, out_zero, out_bias, num_heads):
        self.q, self.k, self.v = q, k, v
        self.out_weight, self.out_des, self.out_bias = out_weight, out_des, out_bias
        self.out_scale = out_scale.reshape(-1).contiguous()
        self._neg_out_zero = (-out_zero).reshape(-1).contiguous()          # kernel convention of the float-input operator
        self.num_heads = int(num_heads)

    @classmethod
    def from_state_dict(cls, sd, prefix="", num_heads=1):
        embed = int(sd[prefix + "q_proj_des"][2])
        bias = sd.get(prefix + "in_proj_bias")

        def proj(name, i):
            qmin, qmax = float(sd[prefix + name + "_quantizer.qmin"]), float(sd[prefix + name + "_quantizer.qmax"])
            a_bits = max(1, (int(round(qmax - qmin))).bit_length())
            return PackedLinear(weight=sd[prefix + name + "_proj_weight"], w_des=sd[prefix + name + "_proj_des"],
                                w_scale=sd[prefix + name + "_proj_scale"], w_zero=sd[prefix + name + "_proj_zero"],
                                bias=None if bias is None else bias[i * embed:(i + 1) * embed].contiguous(),
                                a_scale=sd[prefix + name + "_quantizer.scale"], a_zero=sd[prefix + name + "_quantizer.zero"],
                                a_qmin=qmin, a_qmax=qmax, a_bits=a_bits, a_signed=qmin < 0)
        return cls(proj("q", 0), proj("k", 1), proj("v", 2), sd[prefix + "out_proj.weight"], sd[prefix + "out_proj_des"],
                   sd[prefix + "out_proj_scale"], sd[prefix + "out_proj_zero"], sd.get(prefix + "out_proj.bias"), num_heads)

    def to(self, device):
        for p in (self.q, self.k, self.v):
            p.to(device)
        for name, val in list(vars(self).items()):
            if torch.is_tensor(val):
                setattr(self, name, val.to(device))
        return self

    def __call__(self, query, key, value, route="packed", need_weights=True):
        L, N, E = query.shape
        S = key.shape[0]
        H, d = self.num_heads, E // self.num_heads
        Q = self.q(query, route).reshape(L, N * H, d).transpose(0, 1)      # (N H, L, d), as F.multi_head_attention_forward splits heads
        K = self.k(key, route).reshape(S, N * H, d).transpose(0, 1)
        V = self.v(value, route).reshape(S, N * H, d).transpose(0, 1)
        attn = torch.softmax(torch.bmm(Q * (float(d) ** -0.5), K.transpose(1, 2)), dim=-1)
        ctx = torch.bmm(attn, V).transpose(0, 1).reshape(L * N, E).contiguous()
        out = quantlinear_forward(ctx, (self.out_weight, self.out_des, self.out_scale, self._neg_out_zero), self.out_bias)
        return out.reshape(L, N, E), (attn.reshape(N, H, L, S).mean(dim=1) if need_weights else None)


def from_state_dict(state_dict, conv_geometry=None, num_heads=None):
    """Every packed layer of a model's state_dict -> {module prefix (no trailing dot): PackedConv2d | PackedLinear |
    PackedMultiheadAttention}: what the packing loop of runner/ptq.py:106-114 / runner/qat.py:84-92 leaves behind, ready to
    run on the engine with no reference Python in the process.  A packed conv / linear is recognised by its `w_des` entry
    (6 fields = conv: n_bits, sign, OC, IC, KH, KW; 4 = linear), an attention block by `q_proj_des`.
    conv_geometry: {prefix: (stride, padding)} -- geometry is not part of the state_dict; default stride 1, padding
    (KH - 1) // 2 ("same" for odd kernels).  num_heads: {prefix: heads} (or one int for every attention block)."""
    conv_geometry = conv_geometry or {}
    layers = {}
    for key in state_dict:
        if key.endswith("w_des"):
            prefix = key[:-len("w_des")]
            name = prefix[:-1] if prefix.endswith(".") else prefix
            des = state_dict[key]
            if des.numel() == 6:
                kh = int(des[4])
                stride, padding = conv_geometry.get(name, (1, (kh - 1) // 2))
                layers[name] = PackedConv2d.from_state_dict(state_dict, prefix, stride=stride, padding=padding)
            elif des.numel() == 4:
                layers[name] = PackedLinear.from_state_dict(state_dict, prefix)
            else:
                raise ValueError("%s: a description of %d fields is neither a conv (6) nor a linear (4)" % (key, des.numel()))
        elif key.endswith("q_proj_des"):
            prefix = key[:-len("q_proj_des")]
            name = prefix[:-1] if prefix.endswith(".") else prefix
            heads = num_heads.get(name) if isinstance(num_heads, dict) else num_heads
            if heads is None:
                raise ValueError("%s: num_heads is not part of a state_dict; pass num_heads" % name)
            layers[name] = PackedMultiheadAttention.from_state_dict(state_dict, prefix, heads)
    return layers
