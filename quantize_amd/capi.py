"""ctypes view of the C ABI (include/quant_engine.h) -- what a non-torch host would bind.

Tensors are only used here as device-memory handles (data_ptr) and for the current HIP
stream; every compute call goes straight into libqe_hip.so.  Used by bench.py and by the
`-m gpu` parity tests, which must exercise the C ABI itself and not only the torch module.
"""
import ctypes
import os

from . import loader

_lib = None

QE_OK = 0
DTYPES = {"uint8": 0, "int8": 1, "int16": 2, "int32": 3, "int64": 4,
          "float16": 5, "float32": 6, "float64": 7}

# every symbol include/quant_engine.h declares
SYMBOLS = ["qe_error_string", "qe_last_hip_error", "qe_version", "qe_target_arch", "qe_packed_nbytes",
           "qe_tpack", "qe_tunpack", "qe_quantconv2d_workspace_bytes", "qe_quantconv2d",
           "qe_quantconv2d_float_input", "qe_quantconv2d_path", "qe_quantlinear", "qe_quantlinear_float_input",
           "qe_quantlinear_path", "qe_quantlinear_float_input_path", "qe_global_avgpool", "qe_conv_prepared_bytes", "qe_quantconv2d_prepared_workspace_bytes",
           "qe_conv_prepare", "qe_quantconv2d_prepared", "qe_quantize_pack", "qe_quantconv2d_float_input_workspace_bytes",
           "qe_quantconv2d_float_input_ws", "qe_conv_f32_prepare", "qe_quantconv2d_float_input_prepared",
           "qe_quantconv2d_float_input_path", "qe_quantconv2d_requant_path", "qe_quantconv2d_requant_workspace_bytes",
           "qe_quantconv2d_requant_prepared", "qe_conv_prepared_layout"]


class QeConvShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "IC", "H", "W", "OC", "KH", "KW", "stride", "padding")]


class QeQParam(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("n_bits", ctypes.c_int32), ("sign", ctypes.c_int32),
                ("scale", ctypes.c_void_p), ("zero", ctypes.c_void_p), ("n_param", ctypes.c_int32)]


class QeRequant(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_void_p), ("zero", ctypes.c_void_p), ("n_param", ctypes.c_int32),
                ("qmin", ctypes.c_float), ("qmax", ctypes.c_float), ("n_bits", ctypes.c_int32), ("sign", ctypes.c_int32)]


class QeError(RuntimeError):
    pass


def lib():
    """dlopen libqe_hip.so (no GPU needed to load it) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("QE_LIB") or loader.lib_path()   # QE_LIB: diagnostic builds only (tools/)
    if not os.path.exists(path):
        raise ImportError("libqe_hip.so not built: run `python -m quantize_amd.build`. No fallback exists.")
    L = ctypes.CDLL(path)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.qe_error_string.restype = ctypes.c_char_p
    L.qe_error_string.argtypes = [i32]
    L.qe_last_hip_error.restype = i32
    L.qe_version.restype = ctypes.c_char_p
    L.qe_target_arch.restype = ctypes.c_char_p
    L.qe_packed_nbytes.restype = i64
    L.qe_packed_nbytes.argtypes = [i64, i32]
    L.qe_tpack.restype = i32
    L.qe_tpack.argtypes = [vp, i32, i64, i32, i32, vp, vp, vp]
    L.qe_tunpack.restype = i32
    L.qe_tunpack.argtypes = [vp, i64, i32, i32, vp, vp]
    L.qe_quantconv2d_workspace_bytes.restype = sz
    L.qe_quantconv2d_workspace_bytes.argtypes = [ctypes.POINTER(QeConvShape), i32, i32]
    L.qe_quantconv2d.restype = i32
    L.qe_quantconv2d.argtypes = [ctypes.POINTER(QeQParam), ctypes.POINTER(QeQParam), vp,
                                 ctypes.POINTER(QeConvShape), vp, vp, sz, vp]
    L.qe_quantconv2d_float_input.restype = i32
    L.qe_quantconv2d_float_input.argtypes = [vp, ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape), vp, vp]
    L.qe_quantconv2d_path.restype = i32
    L.qe_quantconv2d_path.argtypes = [ctypes.POINTER(QeConvShape), ctypes.POINTER(QeQParam), ctypes.POINTER(QeQParam)]
    L.qe_quantlinear.restype = i32
    L.qe_quantlinear.argtypes = [ctypes.POINTER(QeQParam), ctypes.POINTER(QeQParam), vp, i64, i32, i32, vp, vp]
    L.qe_quantlinear_float_input.restype = i32
    L.qe_quantlinear_float_input.argtypes = [vp, ctypes.POINTER(QeQParam), vp, i64, i32, i32, vp, vp]
    L.qe_quantlinear_path.restype = i32
    L.qe_quantlinear_path.argtypes = [ctypes.POINTER(QeQParam), ctypes.POINTER(QeQParam), i64, i32, i32]
    L.qe_quantlinear_float_input_path.restype = i32
    L.qe_quantlinear_float_input_path.argtypes = [vp, ctypes.POINTER(QeQParam), i64, i32, i32]
    L.qe_global_avgpool.restype = i32
    L.qe_global_avgpool.argtypes = [vp, i64, i32, vp, vp]
    L.qe_conv_prepared_bytes.restype = sz
    L.qe_conv_prepared_bytes.argtypes = [ctypes.POINTER(QeConvShape), i32, i32]
    L.qe_quantconv2d_prepared_workspace_bytes.restype = sz
    L.qe_quantconv2d_prepared_workspace_bytes.argtypes = [ctypes.POINTER(QeConvShape), i32, i32]
    L.qe_conv_prepare.restype = i32
    L.qe_conv_prepare.argtypes = [ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape), i32, vp, sz, vp]
    L.qe_quantconv2d_prepared.restype = i32
    L.qe_quantconv2d_prepared.argtypes = [ctypes.POINTER(QeQParam), ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape),
                                          vp, sz, vp, vp, sz, vp]
    L.qe_quantconv2d_float_input_workspace_bytes.restype = sz
    L.qe_quantconv2d_float_input_workspace_bytes.argtypes = [ctypes.POINTER(QeConvShape), i32]
    L.qe_quantconv2d_float_input_ws.restype = i32
    L.qe_quantconv2d_float_input_ws.argtypes = [vp, ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape), vp, vp, sz, vp]
    L.qe_conv_f32_prepare.restype = i32
    L.qe_conv_f32_prepare.argtypes = [ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape), vp, sz, vp]
    L.qe_quantconv2d_float_input_prepared.restype = i32
    L.qe_quantconv2d_float_input_prepared.argtypes = [vp, ctypes.POINTER(QeQParam), vp, ctypes.POINTER(QeConvShape), vp, sz, vp, vp]
    L.qe_quantconv2d_float_input_path.restype = i32
    L.qe_quantconv2d_float_input_path.argtypes = [ctypes.POINTER(QeConvShape), ctypes.POINTER(QeQParam)]
    L.qe_quantize_pack.restype = i32
    L.qe_quantize_pack.argtypes = [vp, i64, vp, vp, i32, i64, ctypes.c_float, ctypes.c_float, i32, i32, vp, vp, vp]
    pq, ps, pr = ctypes.POINTER(QeQParam), ctypes.POINTER(QeConvShape), ctypes.POINTER(QeRequant)
    L.qe_quantconv2d_requant_path.restype = i32
    L.qe_quantconv2d_requant_path.argtypes = [ps, pq, pq, pr]
    L.qe_quantconv2d_requant_workspace_bytes.restype = sz
    L.qe_quantconv2d_requant_workspace_bytes.argtypes = [ps, pq, pq, pr]
    L.qe_quantconv2d_requant_prepared.restype = i32
    L.qe_quantconv2d_requant_prepared.argtypes = [pq, pq, vp, ps, vp, sz, pr, vp, vp, vp, sz, vp]
    _lib = L
    return L


def check(rc):
    if rc != QE_OK:
        L = lib()
        msg = L.qe_error_string(rc).decode()
        if rc == 5:
            msg += " (hipError_t %d)" % L.qe_last_hip_error()
        raise QeError(msg)


def _stream(stream=None):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def packed_nbytes(n, n_bits):
    return int(lib().qe_packed_nbytes(int(n), int(n_bits)))


def tpack(x, n_bits, sign, out=None, status=None, stream=None):
    """qe_tpack on a contiguous device tensor. Returns (packed uint8 tensor, status int32[1] tensor)."""
    import torch
    assert x.is_cuda and x.is_contiguous()
    n = x.numel()
    if out is None:
        out = torch.empty(packed_nbytes(n, n_bits), dtype=torch.uint8, device=x.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(lib().qe_tpack(x.data_ptr(), DTYPES[str(x.dtype).replace("torch.", "")], n, int(n_bits),
                         1 if sign else 0, out.data_ptr(), status.data_ptr(), _stream(stream)))
    return out, status


def tunpack(packed, n, n_bits, sign, out=None, stream=None):
    import torch
    assert packed.is_cuda and packed.is_contiguous() and packed.dtype == torch.uint8
    if out is None:
        out = torch.empty(n, dtype=torch.int8 if sign else torch.uint8, device=packed.device)
    check(lib().qe_tunpack(packed.data_ptr(), int(n), int(n_bits), 1 if sign else 0, out.data_ptr(),
                           _stream(stream)))
    return out


def conv_shape(N, IC, H, W, OC, KH, KW, stride, padding):
    return QeConvShape(int(N), int(IC), int(H), int(W), int(OC), int(KH), int(KW), int(stride), int(padding))


def out_hw(sh):
    return ((sh.H + 2 * sh.padding - sh.KH) // sh.stride + 1, (sh.W + 2 * sh.padding - sh.KW) // sh.stride + 1)


def qparam(data, n_bits, sign, scale, zero):
    """Packed operand: uint8 stream + fp32 scale/zero tensors (1 element = per tensor)."""
    assert scale.numel() == zero.numel()
    q = QeQParam(data.data_ptr(), int(n_bits), 1 if sign else 0, scale.data_ptr(), zero.data_ptr(),
                 int(scale.numel()))
    q._keep = (data, scale, zero)  # keep the tensors alive as long as the struct
    return q


def workspace_bytes(sh, x_bits, w_bits):
    return int(lib().qe_quantconv2d_workspace_bytes(ctypes.byref(sh), int(x_bits), int(w_bits)))


def reload_env():
    """Re-read the QE_* tuning knobs: the library snapshots them once per process (qe_common.h env_get); call this after
    changing one inside a live process (tests, A/B tools).  Not part of the public C ABI."""
    lib().qe_debug_reload_env()


def conv_prepared_layout(sh, x_bits, w_bits):
    f = lib().qe_conv_prepared_layout
    f.restype = ctypes.c_uint64
    return int(f(ctypes.byref(sh), int(x_bits), int(w_bits)))


def conv_path(sh, xq, wq):
    return int(lib().qe_quantconv2d_path(ctypes.byref(sh), ctypes.byref(xq), ctypes.byref(wq)))


def quantconv2d(xq, wq, bias, sh, out=None, workspace=None, stream=None):
    import torch
    dev = wq._keep[0].device
    OH, OW = out_hw(sh)
    if out is None:
        out = torch.empty((sh.N, sh.OC, OH, OW), dtype=torch.float32, device=dev)
    need = workspace_bytes(sh, xq.n_bits, wq.n_bits)
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib().qe_quantconv2d(ctypes.byref(xq), ctypes.byref(wq),
                               None if bias is None else bias.data_ptr(), ctypes.byref(sh), out.data_ptr(),
                               None if workspace is None else workspace.data_ptr(),
                               0 if workspace is None else workspace.numel(), _stream(stream)))
    return out


def conv_prepare(wq, bias, sh, x_bits, stream=None):
    """qe_conv_prepare: the x-independent tables of a conv layer, once.  Returns a uint8 device tensor (possibly empty)."""
    import torch
    dev = wq._keep[0].device
    need = int(lib().qe_conv_prepared_bytes(ctypes.byref(sh), int(x_bits), wq.n_bits))
    prepared = torch.empty(max(need, 0), dtype=torch.uint8, device=dev)
    check(lib().qe_conv_prepare(ctypes.byref(wq), None if bias is None else bias.data_ptr(), ctypes.byref(sh), int(x_bits),
                                prepared.data_ptr() if need else None, need, _stream(stream)))
    return prepared


def quantconv2d_prepared(xq, wq, bias, sh, prepared, out=None, workspace=None, stream=None):
    import torch
    dev = wq._keep[0].device
    OH, OW = out_hw(sh)
    if out is None:
        out = torch.empty((sh.N, sh.OC, OH, OW), dtype=torch.float32, device=dev)
    need = int(lib().qe_quantconv2d_prepared_workspace_bytes(ctypes.byref(sh), xq.n_bits, wq.n_bits))
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib().qe_quantconv2d_prepared(ctypes.byref(xq), ctypes.byref(wq), None if bias is None else bias.data_ptr(),
                                        ctypes.byref(sh), prepared.data_ptr() if prepared.numel() else None, prepared.numel(),
                                        out.data_ptr(), None if workspace is None else workspace.data_ptr(),
                                        0 if workspace is None else workspace.numel(), _stream(stream)))
    return out


def requant(scale, zero, qmin, qmax, n_bits, sign):
    """qe_requant: the consumer's activation quantiser (module convention: q = round(y / scale - zero).clamp(qmin, qmax))."""
    r = QeRequant(scale.data_ptr(), zero.data_ptr(), int(scale.numel()), float(qmin), float(qmax), int(n_bits), 1 if sign else 0)
    r._keep = (scale, zero)
    return r


def requant_path(sh, xq, wq, rq):
    return int(lib().qe_quantconv2d_requant_path(ctypes.byref(sh), ctypes.byref(xq), ctypes.byref(wq), ctypes.byref(rq)))


def quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq, out=None, status=None, workspace=None, stream=None):
    """qe_quantconv2d_requant_prepared: conv + the consumer's quantiser + tpack in one call; returns (packed uint8, status)."""
    import torch
    dev = wq._keep[0].device
    OH, OW = out_hw(sh)
    if out is None:
        out = torch.empty(packed_nbytes(sh.N * sh.OC * OH * OW, rq.n_bits), dtype=torch.uint8, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    need = int(lib().qe_quantconv2d_requant_workspace_bytes(ctypes.byref(sh), ctypes.byref(xq), ctypes.byref(wq), ctypes.byref(rq)))
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib().qe_quantconv2d_requant_prepared(ctypes.byref(xq), ctypes.byref(wq), None if bias is None else bias.data_ptr(),
                                                ctypes.byref(sh), prepared.data_ptr() if prepared.numel() else None, prepared.numel(),
                                                ctypes.byref(rq), out.data_ptr(), status.data_ptr(),
                                                None if workspace is None else workspace.data_ptr(),
                                                0 if workspace is None else workspace.numel(), _stream(stream)))
    return out, status


def quantize_pack(x, scale, zero, qmin, qmax, n_bits, sign, inner=1, out=None, status=None, stream=None):
    """qe_quantize_pack: round(x / scale - zero).clamp(qmin, qmax) packed to n_bits; per channel when scale has > 1
    element (channel(i) = (i / inner) % numel(scale))."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and scale.numel() == zero.numel()
    n = x.numel()
    if out is None:
        out = torch.empty(packed_nbytes(n, n_bits), dtype=torch.uint8, device=x.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(lib().qe_quantize_pack(x.data_ptr(), n, scale.data_ptr(), zero.data_ptr(), int(scale.numel()), int(inner),
                                 float(qmin), float(qmax), int(n_bits), 1 if sign else 0, out.data_ptr(), status.data_ptr(),
                                 _stream(stream)))
    return out, status


def quantconv2d_float_input(x, wq, bias, sh, out=None, stream=None, mfma=True):
    """mfma=True: qe_quantconv2d_float_input_ws (bf16 MFMA kernel where eligible); False: the order-preserving VALU kernel."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
    OH, OW = out_hw(sh)
    if out is None:
        out = torch.empty((sh.N, sh.OC, OH, OW), dtype=torch.float32, device=x.device)
    bp = None if bias is None else bias.data_ptr()
    if mfma:
        need = int(lib().qe_quantconv2d_float_input_workspace_bytes(ctypes.byref(sh), wq.n_bits))
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
        check(lib().qe_quantconv2d_float_input_ws(x.data_ptr(), ctypes.byref(wq), bp, ctypes.byref(sh), out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream(stream)))
    else:
        check(lib().qe_quantconv2d_float_input(x.data_ptr(), ctypes.byref(wq), bp, ctypes.byref(sh), out.data_ptr(), _stream(stream)))
    return out


def float_input_path(sh, wq):
    return int(lib().qe_quantconv2d_float_input_path(ctypes.byref(sh), ctypes.byref(wq)))


def conv_f32_prepare(wq, bias, sh, stream=None):
    import torch
    need = int(lib().qe_quantconv2d_float_input_workspace_bytes(ctypes.byref(sh), wq.n_bits))
    prepared = torch.empty(need, dtype=torch.uint8, device=wq._keep[0].device)
    check(lib().qe_conv_f32_prepare(ctypes.byref(wq), None if bias is None else bias.data_ptr(), ctypes.byref(sh),
                                    prepared.data_ptr() if need else None, need, _stream(stream)))
    return prepared


def quantconv2d_float_input_prepared(x, wq, bias, sh, prepared, out=None, stream=None):
    import torch
    OH, OW = out_hw(sh)
    if out is None:
        out = torch.empty((sh.N, sh.OC, OH, OW), dtype=torch.float32, device=x.device)
    check(lib().qe_quantconv2d_float_input_prepared(x.data_ptr(), ctypes.byref(wq), None if bias is None else bias.data_ptr(),
                                                    ctypes.byref(sh), prepared.data_ptr() if prepared.numel() else None,
                                                    prepared.numel(), out.data_ptr(), _stream(stream)))
    return out


def linear_path(xq, wq, B, K, O):
    return int(lib().qe_quantlinear_path(ctypes.byref(xq), ctypes.byref(wq), int(B), int(K), int(O)))


def quantlinear(xq, wq, bias, B, K, O, out=None, stream=None):
    """out[b,o] = bias[o] + sum_k (qx + zx[b]) (qw + zw[o]) sx[b] sw[o]  (the reference kernel's (q + zero) convention)."""
    import torch
    if out is None:
        out = torch.empty((B, O), dtype=torch.float32, device=wq._keep[0].device)
    check(lib().qe_quantlinear(ctypes.byref(xq), ctypes.byref(wq), None if bias is None else bias.data_ptr(),
                               int(B), int(K), int(O), out.data_ptr(), _stream(stream)))
    return out


def linear_float_input_path(x, wq, B, K, O):
    return int(lib().qe_quantlinear_float_input_path(x.data_ptr(), ctypes.byref(wq), int(B), int(K), int(O)))


def quantlinear_float_input(x, wq, bias, O, out=None, stream=None):
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 2
    B, K = x.shape
    if out is None:
        out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    check(lib().qe_quantlinear_float_input(x.data_ptr(), ctypes.byref(wq), None if bias is None else bias.data_ptr(),
                                           int(B), int(K), int(O), out.data_ptr(), _stream(stream)))
    return out


def global_avgpool(x, out=None, stream=None):
    """mean over the last two dims of a contiguous fp32 NCHW tensor -> (N, C)."""
    import torch
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    check(lib().qe_global_avgpool(x.data_ptr(), N * C, H * W, out.data_ptr(), _stream(stream)))
    return out
