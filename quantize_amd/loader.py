"""Locate and import the in-tree native artefacts. No fallbacks: a missing build is an ImportError."""
import importlib.util
import os
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_DIR = os.path.join(_HERE, "_ext")


def lib_path():
    return os.path.join(EXT_DIR, "libqe_hip.so")


def module_path():
    return os.path.join(EXT_DIR, "quant_engine" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


_BUILD_HINT = ("quantize_amd native extension not built: run `python -m quantize_amd.build` "
               "(needs hipcc, cross-compiles gfx950 without a GPU). There is no CPU/Python fallback.")


def load_quant_engine():
    """Import the torch-facing `quant_engine` module and register it as a top-level module, so
    that `from quant_engine import *` (the reference's engine/__init__.py:3) resolves to it."""
    mod = sys.modules.get("quant_engine")
    if mod is not None:
        return mod
    path = module_path()
    if not os.path.exists(path) or not os.path.exists(lib_path()):
        raise ImportError(_BUILD_HINT + " Missing: " + path)
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("quant_engine", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["quant_engine"] = mod
    return mod
