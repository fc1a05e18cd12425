#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only; nothing is written
there: PYTHONDONTWRITEBYTECODE is forced).  The GPU box never runs this script -- it only
reads the committed .npz files.

What is taken from the reference itself:
  G1  tpack/tunpack vectors: produced by the reference's own pure-Python packer
      engine/utils/tensor_packing.py (tpack :11-39, tunpack :42-70), loaded by path.
      That packer writes floor(n*b/8) bytes (:28) and therefore cannot pack a stream whose
      bit length is not a multiple of 8; ragged sizes are produced by padding the input with
      elements whose stored code is 0 (value -2^(b-1) when signed, 0 otherwise) up to a
      multiple of 8 elements and keeping the first ceil(n*b/8) bytes -- element i only ever
      touches bits [i*b, (i+1)*b), so the prefix is what the C++ reference (tpack.cu:154-189)
      writes into its zero-initialised buffer.
  G3  conv vectors: operands packed by the reference packer; expected outputs are
      (a) `ref_fconv`: the reference's packed-forward fallback, F.conv2d on the dequantised
          fp32 tensors (modelzoo/modules/quantconv2d.py:207-210) in the KERNEL sign
          convention (q - zero) * scale (quantconv2d.cu:113-115,128-130),
      (b) `exact64`: the same in float64 (torch, independent of our C code),
      (c) `chain32` / `chain32_fma`: our C oracle's restatement of the CUDA loop
          (regression pin of the oracle itself).
  G4  module capture: the reference's QuantConv2d (modelzoo/modules/quantconv2d.py) is
      driven through calibrate -> pack() -> state_dict -> load_state_dict -> forward on CPU.
      `modelzoo/__init__.py` needs torchvision/robustbench (absent), so `modelzoo.modules`
      is imported under a stub parent package, with an `engine` module that exposes the
      reference's Python tpack/tunpack (the compiled extension does not exist here).

Usage:  python oracle/gen_golden.py   (from the repo root)
"""
import importlib.util
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import oracle  # noqa: E402  (our C restatement, for the chain32 columns)


def load_ref_packer():
    spec = importlib.util.spec_from_file_location(
        "ref_tensor_packing", os.path.join(REF, "engine", "utils", "tensor_packing.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


REFPACK = load_ref_packer()


def ref_pack(x_np, n_bits, sign):
    """Reference packer on an arbitrary-length array (see module docstring for the padding)."""
    x = np.asarray(x_np, dtype=np.float32)
    shape = x.shape
    flat = x.reshape(-1)
    n = flat.size
    pad = (-n) % 8
    fill = -(2 ** (n_bits - 1)) if sign else 0
    padded = np.concatenate([flat, np.full(pad, fill, dtype=np.float32)])
    q, _ = REFPACK.tpack(torch.from_numpy(padded), n_bits, bool(sign))
    nbytes = (n * n_bits + 7) // 8
    packed = q.numpy()[:nbytes].copy()
    des = np.array([n_bits, 1 if sign else 0, *shape], dtype=np.int32)
    return packed, des


def ref_unpack(packed, des):
    """Reference unpacker; pads the byte stream so its Python loop can read whole elements."""
    n_bits = int(des[0])
    n = int(np.prod(des[2:]))
    pad_n = n + ((-n) % 8)
    buf = np.zeros(pad_n * n_bits // 8, dtype=np.uint8)
    buf[:packed.size] = packed
    des_p = torch.tensor([n_bits, int(des[1]), pad_n], dtype=torch.int32)
    u = REFPACK.tunpack(torch.from_numpy(buf), des_p).numpy()[:n]
    return u.reshape([int(v) for v in des[2:]])


def qrange(n_bits, sign):
    return (-(2 ** (n_bits - 1)), 2 ** (n_bits - 1) - 1) if sign else (0, 2 ** n_bits - 1)


# --------------------------------------------------------------------------------------
def gen_g1():
    rng = np.random.RandomState(1234)
    out = {}
    index = []
    shapes = [(8,), (5,), (24,), (13,), (1001,), (16, 8, 3, 3), (64, 3, 7, 7)]
    for b in range(1, 9):
        for sign in (0, 1):
            lo, hi = qrange(b, sign)
            for shape in shapes:
                x = rng.randint(lo, hi + 1, size=shape).astype(np.float32)
                # make sure both range ends occur
                flat = x.reshape(-1)
                flat[0], flat[-1] = lo, hi
                packed, des = ref_pack(x, b, sign)
                unpacked = ref_unpack(packed, des)
                assert np.array_equal(unpacked.astype(np.float32), x), (b, sign, shape)
                key = "b%d_s%d_%s" % (b, sign, "x".join(map(str, shape)))
                out[key + "_x"] = x
                out[key + "_packed"] = packed
                out[key + "_des"] = des
                out[key + "_unpacked"] = unpacked
                index.append(key)
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g1_tpack.npz"), **out)
    print("G1: %d vectors" % len(index))


# --------------------------------------------------------------------------------------
def dequant(q, zero, scale, axis, ndim):
    """(q - zero) * scale in fp32 with per-tensor or per-channel (along `axis`) parameters."""
    q = torch.from_numpy(q.astype(np.float32))
    z = torch.from_numpy(np.asarray(zero, dtype=np.float32))
    s = torch.from_numpy(np.asarray(scale, dtype=np.float32))
    if z.numel() > 1:
        shape = [1] * ndim
        shape[axis] = -1
        z, s = z.view(shape), s.view(shape)
    return (q - z) * s


G3_SHAPES = [
    # (N, IC, H, W, OC, K, stride, pad)
    (2, 3, 9, 9, 4, 3, 1, 1),
    (2, 3, 9, 9, 4, 3, 2, 1),
    (2, 3, 9, 9, 4, 1, 1, 0),
    (2, 3, 9, 9, 4, 1, 2, 0),
    (2, 3, 9, 9, 4, 7, 2, 3),
    (1, 16, 14, 14, 32, 3, 1, 1),
    (2, 32, 7, 7, 48, 1, 1, 0),
    (1, 40, 10, 6, 33, 3, 2, 1),
]

# (w_bits, w_sign, a_bits (0 = fp32 input), a_sign)
G3_QUANT = [(8, 1, 8, 1), (4, 1, 4, 1), (8, 1, 0, 0), (4, 1, 0, 0), (3, 1, 5, 0), (8, 0, 8, 0), (6, 0, 7, 1)]


def gen_g3():
    rng = np.random.RandomState(4321)
    out = {}
    index = []
    case = 0
    for (N, IC, H, W, OC, K, stride, pad) in G3_SHAPES:
        for (wb, wsgn, ab, asgn) in G3_QUANT:
            # rotate the remaining options so the set stays small but covers every value
            w_per_channel = (case % 2) == 0
            a_per_channel = ab != 0 and (case % 3) == 1
            nonzero_zero = (case % 4) in (1, 2)
            with_bias = (case % 2) == 1
            case += 1

            wlo, whi = qrange(wb, wsgn)
            qw = rng.randint(wlo, whi + 1, size=(OC, IC, K, K)).astype(np.int32)
            n_ws = OC if w_per_channel else 1
            sw = rng.uniform(2.5e-4, 7.5e-4, size=n_ws).astype(np.float32)
            if w_per_channel and OC > 1:
                sw[1] = -sw[1]  # BN folding into the scale can make it negative
            zw = (rng.uniform(-3.0, 3.0, size=n_ws).astype(np.float32) if nonzero_zero
                  else np.zeros(n_ws, np.float32))
            w_packed, w_des = ref_pack(qw, wb, wsgn)
            w_scale_t = sw.reshape(-1, 1, 1, 1) if w_per_channel else sw  # arrives as (C,1,1,1)

            bias = rng.normal(0, 0.1, size=OC).astype(np.float32) if with_bias else None
            key = "c%03d" % len(index)
            meta = dict(N=N, IC=IC, H=H, W=W, OC=OC, K=K, stride=stride, pad=pad, wb=wb, wsgn=wsgn,
                        ab=ab, asgn=asgn)
            wf = dequant(qw, zw, sw, 0, 4)

            if ab == 0:
                x = rng.normal(0, 1, size=(N, IC, H, W)).astype(np.float32)
                xf = torch.from_numpy(x)
                out[key + "_x"] = x
                chain = oracle.quantconv2d_float_input(x, w_packed, w_des, sw, zw, bias, stride, pad, mode="fp32")
                chain_fma = oracle.quantconv2d_float_input(x, w_packed, w_des, sw, zw, bias, stride, pad, mode="fp32_fma")
            else:
                alo, ahi = qrange(ab, asgn)
                qx = rng.randint(alo, ahi + 1, size=(N, IC, H, W)).astype(np.int32)
                n_as = IC if a_per_channel else 1
                sx = (rng.uniform(1e-3, 3e-3, size=n_as).astype(np.float32) if a_per_channel
                      else np.array([2e-3], np.float32))
                if nonzero_zero:
                    zx = (rng.uniform(-5.0, 5.0, size=n_as).astype(np.float32) if asgn
                          else rng.uniform(0.3, 0.7, size=n_as).astype(np.float32) * (ahi + 1))
                else:
                    zx = np.zeros(n_as, np.float32)
                x_packed, x_des = ref_pack(qx, ab, asgn)
                xf = dequant(qx, zx, sx, 1, 4)
                out[key + "_x_packed"] = x_packed
                out[key + "_x_des"] = x_des
                out[key + "_x_scale"] = sx
                out[key + "_x_zero"] = zx
                chain = oracle.quantconv2d(x_packed, x_des, sx, zx, w_packed, w_des, sw, zw, bias, stride, pad, mode="fp32")
                chain_fma = oracle.quantconv2d(x_packed, x_des, sx, zx, w_packed, w_des, sw, zw, bias, stride, pad, mode="fp32_fma")

            b_t = None if bias is None else torch.from_numpy(bias)
            # (a) the reference's fallback arithmetic (quantconv2d.py:207-210), fp32
            ref_fconv = F.conv2d(xf, wf, b_t, stride, pad).numpy()
            # (b) float64 ground truth from torch (independent of the C oracle)
            if ab == 0:
                xf64 = torch.from_numpy(x).double()
            else:
                xf64 = dequant_f64(qx, zx, sx, 1)
            wf64 = dequant_f64(qw, zw, sw, 0)
            exact64 = F.conv2d(xf64, wf64, None if bias is None else b_t.double(), stride, pad).numpy()

            out[key + "_w_packed"] = w_packed
            out[key + "_w_des"] = w_des
            out[key + "_w_scale"] = np.asarray(w_scale_t)
            out[key + "_w_zero"] = zw.reshape(np.asarray(w_scale_t).shape)
            if bias is not None:
                out[key + "_bias"] = bias
            out[key + "_stride_pad"] = np.array([stride, pad], np.int32)
            out[key + "_ref_fconv"] = ref_fconv
            out[key + "_exact64"] = exact64
            out[key + "_chain32"] = chain
            out[key + "_chain32_fma"] = chain_fma
            out[key + "_meta"] = np.array([meta[k] for k in
                                           ("N", "IC", "H", "W", "OC", "K", "stride", "pad", "wb", "wsgn", "ab", "asgn")],
                                          np.int32)
            index.append(key)
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g3_conv.npz"), **out)
    print("G3: %d cases" % len(index))


def dequant_f64(q, zero, scale, axis):
    q = torch.from_numpy(q.astype(np.float64))
    z = torch.from_numpy(np.asarray(zero, dtype=np.float32)).double()
    s = torch.from_numpy(np.asarray(scale, dtype=np.float32)).double()
    if z.numel() > 1:
        shape = [1] * 4
        shape[axis] = -1
        z, s = z.view(shape), s.view(shape)
    return (q - z) * s


# --------------------------------------------------------------------------------------
def import_ref_modules():
    """modelzoo.modules under a stub parent package + an `engine` exposing the Python packer."""
    sys.path.insert(0, REF)  # for the reference's top-level `utils` package (Register)
    eng = types.ModuleType("engine")
    eng.tpack, eng.tunpack = REFPACK.tpack, REFPACK.tunpack

    def _absent(*a, **k):
        raise NotImplementedError("compiled reference kernels are not available in this container")

    for name in ("linear", "quantlinear", "quantlinear_float_input", "conv2d", "quantconv2d",
                 "quantconv2d_float_input"):
        setattr(eng, name, _absent)
    sys.modules["engine"] = eng
    pkg = types.ModuleType("modelzoo")
    pkg.__path__ = [os.path.join(REF, "modelzoo")]
    sys.modules["modelzoo"] = pkg
    import modelzoo.modules as mm
    return mm


def gen_g4():
    mm = import_ref_modules()
    out = {}
    index = []
    torch.manual_seed(7)
    cfgs = [
        # name, in, out, k, stride, pad, w_setting, a_setting
        ("w8a8_sym", 16, 8, 3, 1, 1,
         dict(n_bits=8, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
        ("w4a4_sym_s2", 8, 16, 3, 2, 1,
         dict(n_bits=4, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=4, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
        ("w8a8_asym", 16, 8, 1, 1, 0,
         dict(n_bits=8, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"})),
        ("w8_layer_a6", 12, 10, 3, 1, 1,
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"}),
         dict(n_bits=6, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
        # round 2: asymmetric per-channel weights on a padded stride-2 3x3 (border-aware zero terms with
        # non-integer zero points), BatchNorm folded into the weight scale with NEGATIVE gammas (negative w_scale,
        # quantconv2d.py:127-133), BatchNorm folded into the weights of a bias-free conv (:115-126), 4-bit
        # asymmetric activations, W4A4 asymmetric
        ("w8a8_asym_pc_s2", 32, 24, 3, 2, 1,
         dict(n_bits=8, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"}), 14, None),
        ("w8a8_bn_into_scale_neg", 16, 12, 3, 1, 1,
         dict(n_bits=8, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"}), 14, "into_scale"),
        ("w8a8_bn_into_weight_nobias", 16, 12, 1, 1, 0,
         dict(n_bits=8, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"}), 7, "into_weight"),
        ("w8a4_asym", 24, 16, 3, 1, 1,
         dict(n_bits=8, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=4, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"}), 8, None),
        ("w4a4_asym_s2", 16, 20, 3, 2, 1,
         dict(n_bits=4, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=4, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"}), 9, None),
    ]
    for cfg in cfgs:
        (name, cin, cout, k, stride, pad, w_set, a_set) = cfg[:8]
        Hin = cfg[8] if len(cfg) > 8 else 12
        bn_mode = cfg[9] if len(cfg) > 9 else None
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=(bn_mode != "into_weight"))
        x = torch.randn(4, cin, Hin, Hin)
        x = torch.relu(x) if not a_set["symmetric"] else x
        if len(cfg) > 8 and not a_set["symmetric"]:
            x = x + 0.37               # min > 0: a non-zero, non-integer activation zero point (minmax.py:143)
        bn = {}
        if bn_mode is not None:
            gamma = torch.randn(cout)           # about half of the BatchNorm weights negative
            bn = {"weight": gamma, "bias": torch.randn(cout) * 0.1, "running_mean": torch.randn(cout) * 0.1,
                  "running_var": torch.rand(cout) + 0.5, "eps": 1e-5, "into_scale": bn_mode == "into_scale"}

        def make():
            return mm.QuantConv2d(cin, cout, k, stride=stride, padding=pad, w_setting=w_set, a_setting=a_set,
                                  bn_folding={kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in bn.items()},
                                  _parameters={"weight": conv.weight.detach().clone(),
                                               "bias": None if conv.bias is None else conv.bias.detach().clone()})

        m = make()
        with torch.no_grad():
            m.calibrating = True
            m(x)                      # calibrate (quantconv2d.py:141-152)
            m.calibrating = False
            for mod in m.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_sim = m(x)              # fake-quant forward (quantconv2d.py:154-168)
            m.pack()                  # quantconv2d.py:170-196 -> tpack
            sd = {kk: vv.clone() for kk, vv in m.state_dict().items()}
            m2 = make()
            m2.load_state_dict(sd)    # quantconv2d.py:218-235 -> tunpack
            for mod in m2.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_packed = m2(x)          # packed forward fallback (quantconv2d.py:207-210)
            qx, a_scale, a_zero = m2.a_quantizer(x)  # (q, scale, zero) in packed mode (quantizer.py:226)
        key = "m_" + name
        out[key + "_x"] = x.numpy()
        out[key + "_qx"] = qx.numpy()                         # fp32, integer valued
        out[key + "_a_scale"] = a_scale.numpy().reshape(-1)
        out[key + "_a_zero_py"] = a_zero.numpy().reshape(-1)   # PYTHON convention: (q + zero) * scale
        out[key + "_a_bits_sign"] = np.array([a_set["n_bits"], int(qx.min() < 0)], np.int32)
        out[key + "_a_qmin_qmax"] = np.array([float(sd["a_quantizer.qmin"]), float(sd["a_quantizer.qmax"])], np.float32)
        out[key + "_weight_packed"] = sd["weight"].numpy()
        out[key + "_w_des"] = sd["w_des"].numpy()
        out[key + "_w_scale"] = sd["w_scale"].numpy()          # (C,1,1,1) or (1,1,1,1)
        out[key + "_w_zero_py"] = sd["w_zero"].numpy()          # PYTHON convention
        out[key + "_bias"] = sd["bias"].numpy()
        out[key + "_stride_pad"] = np.array([stride, pad], np.int32)
        out[key + "_y_sim"] = y_sim.numpy()
        out[key + "_y_packed"] = y_packed.numpy()
        index.append(key)
        print("G4 %s: max|sim-packed| = %.3g, w_des=%s, w_scale shape %s" % (
            name, float((y_sim - y_packed).abs().max()), sd["w_des"].tolist(), tuple(sd["w_scale"].shape)))
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g4_module.npz"), **out)


# --------------------------------------------------------------------------------------
# G5: quantlinear / quantlinear_float_input cases.  Packed operands from the REFERENCE packer; expected values are
#   (a) the reference module's own packed-forward arithmetic (quantlinear.py:158-161):
#       F.linear((q_x + z_x) * s_x, (q_w + z_w) * s_w, bias) in fp32,
#   (b) the same in float64 (torch), independent of the C oracle,
#   (c) the C oracle's fp32 chains (for the tolerance bookkeeping of the GPU tests).
# Zeros are stored in the convention of the kernel they are passed to: quantlinear takes (q + zero)
# (quantlinear.cu:115,120), quantlinear_float_input takes (q - zero) (quantlinear_float_input.cu:82-86).
G5_SHAPES = [(1, 1, 1), (3, 5, 2), (7, 32, 9), (4, 37, 33), (33, 64, 65), (16, 160, 40), (5, 768, 24)]
G5_QUANT = [(8, 1, 8, 1), (8, 0, 8, 0), (4, 1, 4, 1), (3, 1, 6, 0), (8, 1, 0, 0), (5, 0, 0, 0)]


def gen_g5():
    rng = np.random.RandomState(2468)
    out, index = {}, []
    case = 0
    for (B, K, O) in G5_SHAPES:
        for (wb, wsgn, ab, asgn) in G5_QUANT:
            w_per_channel = (case % 2) == 0
            a_per_row = ab != 0 and (case % 3) == 1
            nonzero_zero = (case % 4) in (1, 2)
            with_bias = (case % 2) == 1
            case += 1
            wlo, whi = qrange(wb, wsgn)
            qw = rng.randint(wlo, whi + 1, size=(O, K)).astype(np.int32)
            n_ws = O if w_per_channel else 1
            sw = rng.uniform(2.5e-4, 7.5e-4, size=n_ws).astype(np.float32)
            zw = (rng.uniform(-3.0, 3.0, size=n_ws).astype(np.float32) if nonzero_zero else np.zeros(n_ws, np.float32))
            w_packed, w_des = ref_pack(qw, wb, wsgn)
            bias = rng.normal(0, 0.1, size=O).astype(np.float32) if with_bias else None
            key = "l%03d" % len(index)
            b_t = None if bias is None else torch.from_numpy(bias)
            qw64 = torch.from_numpy(qw.astype(np.float64))
            sw64 = torch.from_numpy(sw).double().view(-1, 1)
            zw64 = torch.from_numpy(zw).double().view(-1, 1)
            if ab == 0:
                x = rng.normal(0, 1, size=(B, K)).astype(np.float32)
                out[key + "_x"] = x
                wf64 = (qw64 - zw64) * sw64                            # (q - zero) * scale
                exact64 = F.linear(torch.from_numpy(x).double(), wf64, None if bias is None else b_t.double()).numpy()
                ref_f = F.linear(torch.from_numpy(x), wf64.float(), b_t).numpy()
                chain = oracle.quantlinear_float_input(x, w_packed, w_des, sw, zw, bias, mode="fp32")
                chain_fma = oracle.quantlinear_float_input(x, w_packed, w_des, sw, zw, bias, mode="fp32_fma")
            else:
                alo, ahi = qrange(ab, asgn)
                qx = rng.randint(alo, ahi + 1, size=(B, K)).astype(np.int32)
                n_as = B if a_per_row else 1
                sx = (rng.uniform(1e-3, 3e-3, size=n_as).astype(np.float32) if a_per_row else np.array([2e-3], np.float32))
                zx = (rng.uniform(-5.0, 5.0, size=n_as).astype(np.float32) if nonzero_zero else np.zeros(n_as, np.float32))
                x_packed, x_des = ref_pack(qx, ab, asgn)
                out[key + "_x_packed"] = x_packed
                out[key + "_x_des"] = x_des
                out[key + "_x_scale"] = sx
                out[key + "_x_zero"] = zx
                xf64 = (torch.from_numpy(qx.astype(np.float64)) + torch.from_numpy(zx).double().view(-1, 1)) * \
                    torch.from_numpy(sx).double().view(-1, 1)          # (q + zero) * scale, per ROW
                wf64 = (qw64 + zw64) * sw64
                exact64 = F.linear(xf64, wf64, None if bias is None else b_t.double()).numpy()
                ref_f = F.linear(xf64.float(), wf64.float(), b_t).numpy()   # quantlinear.py:158-161
                chain = oracle.quantlinear(x_packed, x_des, sx, zx, w_packed, w_des, sw, zw, bias, mode="fp32")
                chain_fma = oracle.quantlinear(x_packed, x_des, sx, zx, w_packed, w_des, sw, zw, bias, mode="fp32_fma")
            out[key + "_w_packed"] = w_packed
            out[key + "_w_des"] = w_des
            out[key + "_w_scale"] = sw
            out[key + "_w_zero"] = zw
            if bias is not None:
                out[key + "_bias"] = bias
            out[key + "_ref_flinear"] = ref_f
            out[key + "_exact64"] = exact64
            out[key + "_chain32"] = chain
            out[key + "_chain32_fma"] = chain_fma
            out[key + "_meta"] = np.array([B, K, O, wb, wsgn, ab, asgn], np.int32)
            index.append(key)
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g5_linear.npz"), **out)
    print("G5: %d cases" % len(index))


def gen_g6():
    """Module capture: the reference's QuantLinear (modelzoo/modules/quantlinear.py) calibrated, packed
    (:123-148 -> tpack), reloaded (:166-186 -> tunpack) and run through its packed forward (:150-161)."""
    mm = import_ref_modules()
    out, index = {}, []
    torch.manual_seed(11)
    cfgs = [
        ("w8a8_sym", 48, 20,
         dict(n_bits=8, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
        ("w8a8_asym", 40, 24,
         dict(n_bits=8, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"})),
        ("w4_layer_a8", 64, 10,
         dict(n_bits=4, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
    ]
    for (name, fin, fout, w_set, a_set) in cfgs:
        lin = torch.nn.Linear(fin, fout, bias=True)
        x = torch.randn(12, fin)
        x = torch.relu(x) if not a_set["symmetric"] else x

        def make():
            return mm.QuantLinear(fin, fout, w_setting=w_set, a_setting=a_set,
                                  _parameters={"weight": lin.weight.detach().clone(), "bias": lin.bias.detach().clone()})

        m = make()
        with torch.no_grad():
            m.calibrating = True
            m(x)
            m.calibrating = False
            for mod in m.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_sim = m(x)
            m.pack()
            sd = {kk: vv.clone() for kk, vv in m.state_dict().items()}
            m2 = make()
            m2.load_state_dict(sd)
            for mod in m2.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_packed = m2(x)
            qx, a_scale, a_zero = m2.a_quantizer(x)
        key = "m_" + name
        out[key + "_x"] = x.numpy()
        out[key + "_qx"] = qx.numpy()
        out[key + "_a_scale"] = a_scale.numpy().reshape(-1)
        out[key + "_a_zero_py"] = a_zero.numpy().reshape(-1)    # PYTHON convention (q + zero) * scale
        out[key + "_a_bits_sign"] = np.array([a_set["n_bits"], int(qx.min() < 0)], np.int32)
        out[key + "_a_qmin_qmax"] = np.array([float(sd["a_quantizer.qmin"]), float(sd["a_quantizer.qmax"])], np.float32)
        out[key + "_weight_packed"] = sd["weight"].numpy()
        out[key + "_w_des"] = sd["w_des"].numpy()
        out[key + "_w_scale"] = sd["w_scale"].numpy()
        out[key + "_w_zero_py"] = sd["w_zero"].numpy()
        out[key + "_bias"] = sd["bias"].numpy()
        out[key + "_y_sim"] = y_sim.numpy()
        out[key + "_y_packed"] = y_packed.numpy()
        index.append(key)
        print("G6 %s: max|sim-packed| = %.3g, w_des=%s, w_scale shape %s, a_scale shape %s" % (
            name, float((y_sim - y_packed).abs().max()), sd["w_des"].tolist(), tuple(sd["w_scale"].shape),
            tuple(a_scale.shape)))
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g6_linear_module.npz"), **out)


def gen_g7():
    """Module capture: the reference's QuantMultiheadAttention (modelzoo/modules/quantmultiheadattention.py) calibrated,
    packed (:165-223: q / k / v projection weights and out_proj through tpack), reloaded (:405-432 -> tunpack) and run
    through its packed forward (:262-...).  Only the separate-projection form (kdim != embed_dim) can be captured: with
    kdim == embed_dim the reference's own pack() dereferences the None q_proj_weight (AttributeError) -- the ViT blocks
    therefore have no reference capture, their projections are covered by the QuantLinear fixtures (G5 / G6)."""
    mm = import_ref_modules()
    out, index = {}, []
    torch.manual_seed(17)
    cfgs = [
        ("w8a8_sym", 32, 4, 24, 5, 7, 3,
         dict(n_bits=8, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
        ("w8a8_asym", 48, 6, 40, 6, 6, 2,
         dict(n_bits=8, symmetric=False, signed=False, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=False, signed=False, granularity="layer", range={"name": "minmax"})),
        ("w4a8_sym", 64, 8, 48, 4, 9, 2,
         dict(n_bits=4, symmetric=True, signed=True, granularity="channel", range={"name": "minmax"}),
         dict(n_bits=8, symmetric=True, signed=True, granularity="layer", range={"name": "minmax"})),
    ]
    for (name, E, H, KD, L, S, N, w_set, a_set) in cfgs:
        ref = torch.nn.MultiheadAttention(E, H, kdim=KD, vdim=KD, bias=True)

        def make():
            return mm.QuantMultiheadAttention(
                E, H, kdim=KD, vdim=KD, w_setting=dict(w_set), a_setting=dict(a_set),
                _parameters={k: (v.detach().clone() if v is not None else None) for k, v in ref._parameters.items()},
                _modules={"out_proj": ref.out_proj})

        q, k, v = torch.randn(L, N, E), torch.randn(S, N, KD), torch.randn(S, N, KD)
        if not a_set["symmetric"]:
            q, k, v = torch.relu(q), torch.relu(k), torch.relu(v)
        m = make()
        with torch.no_grad():
            m.calibrating = True
            m(q, k, v)
            m.calibrating = False
            for mod in m.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_sim, _ = m(q, k, v)
            m.pack()
            sd = {kk: vv.clone() for kk, vv in m.state_dict().items()}
            m2 = make()
            m2.load_state_dict(sd)
            for mod in m2.modules():
                if isinstance(mod, mm.Quantizer):
                    mod.quant(True)
            y_packed, attn = m2(q, k, v)
        key = "m_" + name
        out[key + "_query"], out[key + "_key"], out[key + "_value"] = q.numpy(), k.numpy(), v.numpy()
        out[key + "_heads"] = np.array([E, H, KD], np.int32)
        for kk, vv in sd.items():
            out[key + "_sd_" + kk] = vv.numpy()
        out[key + "_y_sim"] = y_sim.numpy()
        out[key + "_y_packed"] = y_packed.numpy()
        out[key + "_attn"] = attn.numpy()
        index.append(key)
        print("G7 %s: max|sim-packed| = %.3g, q_proj_des=%s, out_proj_des=%s" % (
            name, float((y_sim - y_packed).abs().max()), sd["q_proj_des"].tolist(), sd["out_proj_des"].tolist()))
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "g7_mha_module.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-g7" in sys.argv:      # added in round 3: leaves the earlier fixtures byte for byte as committed
        gen_g7()
        sys.exit(0)
    gen_g1()
    gen_g3()
    gen_g4()
    gen_g5()
    gen_g6()
    gen_g7()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
