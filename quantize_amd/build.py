"""Build the native parts of quantize_amd in-tree.

  quantize_amd/_ext/libqe_hip.so                     hipcc, --offload-arch=gfx950:
                                                     HIP kernels + the C ABI of include/quant_engine.h
  quantize_amd/_ext/quant_engine.cpython-*.so        g++: the torch-facing pybind module (csrc/torch_binding.cpp),
                                                     linked against libqe_hip.so ($ORIGIN rpath)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container;
the products are git-ignored and travel to the GPU box with the gpurun snapshot.
Run as `python -m quantize_amd.build [--force]`.
"""
import os
import subprocess
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
EXT_DIR = os.path.join(_HERE, "_ext")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

HIP_SOURCES = ["qe_api.hip", "qe_tpack.hip", "qe_conv_generic.hip", "qe_conv_mfma.hip",
               "qe_conv_mfma_i0.hip", "qe_conv_mfma_i1.hip", "qe_conv_mfma_i2.hip", "qe_conv_mfma_i3.hip", "qe_conv_mfma_i4.hip", "qe_conv_mfma_i5.hip", "qe_conv_mfma_i6.hip", "qe_conv_mfma_i7.hip", "qe_conv_mfma_i8.hip", "qe_linear.hip", "qe_conv_flatd.hip", "qe_conv_f32.hip", "qe_conv_pwr.hip"]
HIP_HEADERS = ["qe_common.h", "qe_conv_mfma_kernel.hpp", os.path.join(INCLUDE, "quant_engine.h")]
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_sha16():
    """Fingerprint of the kernel sources (csrc/* + include/quant_engine.h).  profiles/traffic.json records the
    fingerprint its PMC passes were measured at; bench.py reports `roofline.traffic` only while it still matches
    (the GPU box has no .git, so a commit id cannot be checked there)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h", ".cpp")))
    for f in files + [os.path.join(INCLUDE, "quant_engine.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lib_path():
    return os.path.join(EXT_DIR, "libqe_hip.so")


def module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(EXT_DIR, "quant_engine" + suffix)


def build_hip(force=False, verbose=False, stamp=False):
    """Each .hip translation unit -> object (in parallel: the MFMA instantiation units dominate),
    then one link into libqe_hip.so.  stamp=True builds the diagnostic variant libqe_hip_stamp.so
    (-DQE_STAMP: in-kernel s_memtime phase stamps; never the library the product loads)."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(EXT_DIR, exist_ok=True)
    obj_dir = os.path.join(EXT_DIR, "obj_stamp" if stamp else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
              "-I", INCLUDE]
    if stamp:
        common.append("-DQE_STAMP=1")
    jobs, objs = [], []
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _newer(op, [sp] + hdrs):
            jobs.append(common + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(run, jobs))
    out = os.path.join(EXT_DIR, "libqe_hip_stamp.so") if stamp else lib_path()
    if jobs or not os.path.exists(out):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_torch_module(force=False, verbose=False):
    import torch
    from torch.utils import cpp_extension

    os.makedirs(EXT_DIR, exist_ok=True)
    src = os.path.join(CSRC, "torch_binding.cpp")
    out = module_path()
    if not force and not _newer(out, [src, os.path.join(INCLUDE, "quant_engine.h"), lib_path()]):
        return out
    incs = cpp_extension.include_paths("cuda") + [sysconfig.get_paths()["include"], INCLUDE]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-Wno-deprecated-declarations",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
           "-DTORCH_EXTENSION_NAME=quant_engine", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for i in incs:
        cmd += ["-isystem", i]
    cmd += [src, "-o", out, "-L", EXT_DIR, "-lqe_hip", "-L", torch_lib,
            "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_torch_module(force, verbose)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--stamp" in sys.argv:
        print("built", build_hip(force=force, verbose=True, stamp=True))
        sys.exit(0)
    for p in build_all(force=force, verbose=True):
        print("built", p)
