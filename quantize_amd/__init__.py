"""quantize_amd -- MI355X (gfx950) native replacement for the `engine.kernels`
extension of JingInAI/Quantize: bit-exact tensor packing and quantized conv2d.

Layout (only what the hot path needs):
  csrc/       hand-written HIP kernels + the C ABI (include/quant_engine.h) + the torch binding
  _ext/       built artefacts (libqe_hip.so, quant_engine.*.so); git-ignored, built by build.py
  loader.py   imports the built `quant_engine` module and registers it under that top-level name
  engine/     mirror of the reference's `engine` facade (engine/__init__.py:1-5)
  operator/   mirror of modelzoo/modules/operator (quantconv2d_forward, QuantConv2dOp1/2, ...)
  capi.py     ctypes view of the C ABI for benches/tests that bypass torch's dispatcher
  resnet50.py the ResNet-50 conv stack of SURVEY.md section 8d (shape table + synthetic inputs)
  dist.py     batch sharding across ranks + all-gather of logits (torch.distributed / RCCL)

Importing this package does not load native code; `quantize_amd.engine` (or
`loader.load_quant_engine()`) does, and raises ImportError if it has not been built --
there is no Python or CPU fallback.
"""
__version__ = "0.1.0"
