"""turbodiffusion_amd — MI355X-native (gfx950) TurboDiffusion denoising hot path.

``ops``  mirrors ``turbodiffusion.ops``  (int8_quant, int8_linear, rmsnorm, layernorm, Int8Linear,
         FastRMSNorm, FastLayerNorm)
``sla``  mirrors ``turbodiffusion.SLA``  (SparseLinearAttention, SageSparseLinearAttention)
``wan``  mirrors the ``WanModel.forward`` surface of ``rcm/networks/wan2pt1.py`` / ``wan2pt2.py``
Everything computes through libturbodiffusion_amd.so (hand-written HIP behind a C ABI).
"""
__version__ = "0.1.0"
