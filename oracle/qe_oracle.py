"""ctypes/numpy front-end of oracle/qe_oracle.c (see that file's header).

TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and bench.py's
cpu_baseline.  Function names and argument meaning follow the reference's
`quant_engine` exports (engine/kernels/pybind.cpp:9-16) but work on numpy
arrays: a packed tensor is (uint8 1-D array, des int32 array) exactly as the
reference's `tpack` returns it (engine/kernels/tpack/tpack.cu:228-254).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libqe_oracle.so")
_lib = None


class OracleError(RuntimeError):
    """Mirrors the reference's TORCH_CHECK -> RuntimeError (tpack.cu:13-14)."""


_ERRORS = {
    1: "n_bits must be in the range (0, 8]",   # tpack.cu:13
    2: "The input tensor is out of range.",    # tpack.cu:14
}


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile). Idempotent."""
    src = os.path.join(_HERE, "qe_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and (not os.path.exists(src) or os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    f32p = ctypes.POINTER(ctypes.c_float)
    f64p = ctypes.POINTER(ctypes.c_double)
    i64, i32 = ctypes.c_int64, ctypes.c_int
    lib.qe_oracle_tpack.argtypes = [f32p, i64, i32, i32, u8p]
    lib.qe_oracle_tpack.restype = i32
    lib.qe_oracle_tunpack.argtypes = [u8p, i64, i32, i32, u8p]
    lib.qe_oracle_tunpack.restype = i32
    lib.qe_oracle_quantconv2d.argtypes = (
        [u8p, i32, i32, f32p, f32p, i32, u8p, i32, i32, f32p, f32p, i32, f32p]
        + [i32] * 9 + [i32, f32p, f64p])
    lib.qe_oracle_quantconv2d.restype = i32
    lib.qe_oracle_quantconv2d_float_input.argtypes = (
        [f32p, u8p, i32, i32, f32p, f32p, i32, f32p] + [i32] * 9 + [i32, f32p, f64p])
    lib.qe_oracle_quantconv2d_float_input.restype = i32
    lib.qe_oracle_quantlinear.argtypes = (
        [u8p, i32, i32, f32p, f32p, i32, u8p, i32, i32, f32p, f32p, i32, f32p] + [i32] * 3 + [i32, f32p, f64p])
    lib.qe_oracle_quantlinear.restype = i32
    lib.qe_oracle_quantlinear_float_input.argtypes = (
        [f32p, u8p, i32, i32, f32p, f32p, i32, f32p] + [i32] * 3 + [i32, f32p, f64p])
    lib.qe_oracle_quantlinear_float_input.restype = i32
    lib.qe_oracle_num_threads.restype = i32
    lib.qe_oracle_set_num_threads.argtypes = [i32]
    _lib = lib
    return lib


def _ptr(a, ty):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _check(rc):
    if rc != 0:
        raise OracleError(_ERRORS.get(rc, "oracle error %d" % rc))


def num_threads():
    return int(_load().qe_oracle_num_threads())


def set_num_threads(n):
    _load().qe_oracle_set_num_threads(int(n))


def tpack(x, n_bits, sign):
    """reference: tpack(x, n_bits, sign) -> [uint8 1-D, des int32]  (tpack.cu:203-255)."""
    lib = _load()
    x = np.asarray(x)
    shape = x.shape
    xf = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)  # x[i].item<float>(), tpack.cu:157
    n = xf.size
    if n == 0:
        # x.min() on an empty tensor raises in torch before any packing happens.
        raise OracleError("min(): Expected reduction dim to be specified for input.numel() == 0.")
    nb = int(n_bits)
    out = np.zeros(((n * nb + 7) // 8) if 0 < nb <= 8 else 1, dtype=np.uint8)
    _check(lib.qe_oracle_tpack(_ptr(xf, ctypes.c_float), n, nb, 1 if sign else 0,
                               _ptr(out, ctypes.c_uint8)))
    des = np.array([nb, 1 if sign else 0, *shape], dtype=np.int32)  # tpack.cu:228-238
    return out, des


def tunpack(packed, des):
    """reference: tunpack(x, des) -> int8/uint8 tensor of shape des[2:]  (tpack.cu:429-476)."""
    lib = _load()
    des = np.asarray(des)
    if des.shape[0] < 3:
        raise OracleError("The description is too short, which should be at least 3.")  # tpack.cu:15,434
    packed = np.asarray(packed)
    if packed.dtype != np.uint8:
        raise OracleError("The input tensor must be torch.uint8.")  # tpack.cu:440
    packed = np.ascontiguousarray(packed).reshape(-1)
    nb, sign = int(des[0]), int(des[1])
    shape = [int(v) for v in des[2:]]
    n = int(np.prod(shape, dtype=np.int64))
    out = np.zeros(n, dtype=np.uint8)
    _check(lib.qe_oracle_tunpack(_ptr(packed, ctypes.c_uint8), n, nb, 1 if sign else 0,
                                 _ptr(out, ctypes.c_uint8)))
    if sign:
        out = out.view(np.int8)  # tpack.cu:453-455
    return out.reshape(shape)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


MODES = {"fp32": 0, "fp32_fma": 1, "f64": 2}


def quantconv2d(x, x_des, x_scale, x_zero, w, w_des, w_scale, w_zero, bias, stride, padding,
                mode="fp32", return_f64=False):
    """reference: quantconv2d(...) 11 positional args (functions/quantconv2d.cu:164-175)."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    w = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1)
    x_des = np.asarray(x_des)
    w_des = np.asarray(w_des)
    xb, xs = int(x_des[0]), int(x_des[1])
    N, IC, H, W = [int(v) for v in x_des[2:6]]       # quantconv2d.cu:193,199-202
    wb, ws = int(w_des[0]), int(w_des[1])
    OC, KH, KW = int(w_des[2]), int(w_des[4]), int(w_des[5])  # :205-207 (weight_shape[1] unread)
    sx, zx, sw, zw = _f32(x_scale), _f32(x_zero), _f32(w_scale), _f32(w_zero)
    b = None if bias is None else _f32(bias)
    OH = (H + 2 * padding - KH) // stride + 1
    OW = (W + 2 * padding - KW) // stride + 1
    out = np.zeros((N, OC, max(OH, 0), max(OW, 0)), dtype=np.float32)
    out64 = np.zeros(out.shape, dtype=np.float64) if return_f64 else None
    _check(lib.qe_oracle_quantconv2d(
        _ptr(x, ctypes.c_uint8), xb, xs, _ptr(sx, ctypes.c_float), _ptr(zx, ctypes.c_float),
        1 if sx.size == 1 else 0,                     # :238 per-tensor iff numel()==1
        _ptr(w, ctypes.c_uint8), wb, ws, _ptr(sw, ctypes.c_float), _ptr(zw, ctypes.c_float),
        1 if sw.size == 1 else 0,                     # :244
        _ptr(b, ctypes.c_float), N, IC, H, W, OC, KH, KW, int(stride), int(padding),
        MODES[mode], _ptr(out, ctypes.c_float), _ptr(out64, ctypes.c_double)))
    return (out, out64) if return_f64 else out


def quantconv2d_float_input(x, w, w_des, w_scale, w_zero, bias, stride, padding,
                            mode="fp32", return_f64=False):
    """reference: quantconv2d_float_input(...) 8 positional args
    (functions/quantconv2d_float_input.cu:140-148)."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    N, IC, H, W = x.shape
    w = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1)
    w_des = np.asarray(w_des)
    wb, ws = int(w_des[0]), int(w_des[1])
    OC, KH, KW = int(w_des[2]), int(w_des[4]), int(w_des[5])
    sw, zw = _f32(w_scale), _f32(w_zero)
    b = None if bias is None else _f32(bias)
    OH = (H + 2 * padding - KH) // stride + 1
    OW = (W + 2 * padding - KW) // stride + 1
    out = np.zeros((N, OC, max(OH, 0), max(OW, 0)), dtype=np.float32)
    out64 = np.zeros(out.shape, dtype=np.float64) if return_f64 else None
    _check(lib.qe_oracle_quantconv2d_float_input(
        _ptr(x, ctypes.c_float), _ptr(w, ctypes.c_uint8), wb, ws,
        _ptr(sw, ctypes.c_float), _ptr(zw, ctypes.c_float), 1 if sw.size == 1 else 0,
        _ptr(b, ctypes.c_float), N, IC, H, W, OC, KH, KW, int(stride), int(padding),
        MODES[mode], _ptr(out, ctypes.c_float), _ptr(out64, ctypes.c_double)))
    return (out, out64) if return_f64 else out


def quantlinear(x, x_des, x_scale, x_zero, w, w_des, w_scale, w_zero, bias, mode="fp32", return_f64=False):
    """reference: quantlinear(...) 9 positional args (functions/quantlinear.cu:233-243).
    KERNEL convention of this op: (q + zero), input scale/zero per batch ROW, weight's per output COLUMN."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    w = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1)
    x_des, w_des = np.asarray(x_des), np.asarray(w_des)
    xb, xs = int(x_des[0]), int(x_des[1])
    B, K = int(x_des[2]), int(x_des[3])               # quantlinear.cu:255,272-273
    wb, ws = int(w_des[0]), int(w_des[1])
    O, K2 = int(w_des[2]), int(w_des[3])              # :258,274
    if K != K2:
        raise OracleError("Input and weight do not match")   # :259
    sx, zx, sw, zw = _f32(x_scale), _f32(x_zero), _f32(w_scale), _f32(w_zero)
    b = None if bias is None else _f32(bias)
    out = np.zeros((B, O), dtype=np.float32)
    out64 = np.zeros(out.shape, dtype=np.float64) if return_f64 else None
    _check(lib.qe_oracle_quantlinear(
        _ptr(x, ctypes.c_uint8), xb, xs, _ptr(sx, ctypes.c_float), _ptr(zx, ctypes.c_float), 1 if sx.size == 1 else 0,
        _ptr(w, ctypes.c_uint8), wb, ws, _ptr(sw, ctypes.c_float), _ptr(zw, ctypes.c_float), 1 if sw.size == 1 else 0,
        _ptr(b, ctypes.c_float), B, K, O, MODES[mode], _ptr(out, ctypes.c_float), _ptr(out64, ctypes.c_double)))
    return (out, out64) if return_f64 else out


def quantlinear_float_input(x, w, w_des, w_scale, w_zero, bias, mode="fp32", return_f64=False):
    """reference: quantlinear_float_input(...) 6 positional args (functions/quantlinear_float_input.cu:120-126).
    (q - zero) * scale convention for the weight."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, K = x.shape
    w = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1)
    w_des = np.asarray(w_des)
    wb, ws = int(w_des[0]), int(w_des[1])
    O = int(w_des[2])                                 # :150 (weight_shape[1] is never compared with K)
    sw, zw = _f32(w_scale), _f32(w_zero)
    b = None if bias is None else _f32(bias)
    out = np.zeros((B, O), dtype=np.float32)
    out64 = np.zeros(out.shape, dtype=np.float64) if return_f64 else None
    _check(lib.qe_oracle_quantlinear_float_input(
        _ptr(x, ctypes.c_float), _ptr(w, ctypes.c_uint8), wb, ws, _ptr(sw, ctypes.c_float), _ptr(zw, ctypes.c_float),
        1 if sw.size == 1 else 0, _ptr(b, ctypes.c_float), B, K, O, MODES[mode],
        _ptr(out, ctypes.c_float), _ptr(out64, ctypes.c_double)))
    return (out, out64) if return_f64 else out
