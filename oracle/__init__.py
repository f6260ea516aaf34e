"""CPU oracle for the MI355X TurboDiffusion denoising path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  ``turbodiffusion_amd`` never imports this package; its operators
raise when the HIP library is missing.

Every function here restates, on the CPU (torch-CPU / numpy, fp32 or exact
integer arithmetic), one operator of the reference hot path and cites the
reference ``file:line`` it follows (paths relative to ``/root/reference``).

Parity pinning status:
  * Wan DiT eager path, norms, AdaLN glue, RoPE: pinned bit-exactly against
    the reference itself, imported on CPU by ``oracle/ref_harness.py``
    (``tests/test_oracle_cpu.py``; fixtures ``tests/golden/wan_tiny.pt``).
  * SLA / SageSLA module compositions (``SparseLinearAttention.forward``,
    ``SageSparseLinearAttention.forward`` FP16-PV and FP8-PV branches,
    ``get_block_map``, the linear branch, ``proj_l`` under autocast): pinned
    bit-exactly against the reference's own module code run on the CPU with
    only its Triton / SpargeAttn / CUDA leaves replaced
    (``ref_harness.patched_sla``; ``tests/golden/sla_tiny.pt`` carries the
    reference-produced tensors to the GPU box).
  * The reference's TRITON leaves — ``_attn_fwd`` (SLA/kernel.py:21-82), ``compress_kernel`` / ``mean_pool`` and
    ``get_block_map`` (SLA/utils.py:21-67), ``rmsnorm`` / ``layernorm`` (ops/core.py:96-136, 193-335), and the whole
    unpatched ``SparseLinearAttention`` module — were EXECUTED on an MI355X (``oracle/triton_leaves.py``: stage, gpurun,
    unstage; Triton 3.6 builds them for gfx950 unmodified) and their outputs are the fixture
    ``tests/golden/triton_leaves.pt``: pooling and block map agree bit for bit, attention to one bf16 rounding step on
    0.05 % of the values, RMSNorm to 3e-7, and LayerNorm once the oracle learnt that the reference sums (x - mean)^2 over
    next_power_of_2(N) columns (``ops_ref.layernorm_fast``).  The Wan2.2 network (``wan2pt2``) is pinned live as well.
  * Block-128 INT8 quantiser and W8A8 GEMM (CUDA sources, not buildable here —
    need nvcc + the un-vendored CUTLASS submodule): restated from the source
    lines cited; the reference ships no test vectors for them.
  * SageAttention INT8-QK / FP16-PV arithmetic lives in the un-vendored
    SpargeAttn dependency (``spas_sage_attn``; the only pin in the tree is commit
    ae5b629e in TurboT2AV/LTX-2/scripts/install_acceleration.sh:5-6):
    **parity unpinned** — the oracle states its own rounding rule
    (``sage_ref.quant_per_block_int8``) and the HIP path is bit-exact to *that*.
"""
