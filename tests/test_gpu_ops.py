"""-m gpu parity tests: HIP kernels (through the C-ABI) vs the CPU oracle — linear/norm operators."""
import pytest
import torch

from oracle import ops_ref as O
from tests.util import act_like, rel_l2, ulp_diff_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from turbodiffusion_amd import kernels
    return kernels


@pytest.mark.parametrize("m,n", [(128, 128), (1, 8), (300, 256), (1000, 1536), (257, 8960 // 4), (5, 136)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_quant_bit_exact(K, m, n, dtype):
    x = act_like(m, n, dtype, seed=m * 7 + n)
    q_ref, s_ref = O.quant_block128(x)
    q, s = K.quant_i8_block128(x.to(DEV))
    assert torch.equal(s.cpu(), s_ref), "scales must be bit-exact"
    assert torch.equal(q.cpu(), q_ref), "int8 codes must be bit-exact"


def test_quant_edge_cases(K):
    # all-zero block -> amax clamps to 1e-8; saturation: +amax -> 127, -amax -> -128
    x = torch.zeros(256, 256, dtype=torch.bfloat16)
    x[130, 5] = 3.0
    x[131, 6] = -3.0
    q, s = K.quant_i8_block128(x.to(DEV))
    q_ref, s_ref = O.quant_block128(x)
    assert torch.equal(q.cpu(), q_ref) and torch.equal(s.cpu(), s_ref)
    assert q[130, 5].item() == 127 and q[131, 6].item() == -128
    assert s[0, 0].item() == pytest.approx(1e-8 / 128, rel=1e-6)


def test_quant_rejects_bad_shape(K):
    from turbodiffusion_amd._lib import TurboDiffusionAMDError
    with pytest.raises(TurboDiffusionAMDError):
        K.quant_i8_block128(torch.zeros(4, 12, dtype=torch.bfloat16, device=DEV))  # n % 8 != 0


@pytest.mark.parametrize("m,n,k", [(128, 128, 128), (200, 384, 512), (1000, 1536, 1536), (77, 136, 256),
                                   (512, 256, 8960 // 2 - 8960 // 2 % 128)])
@pytest.mark.parametrize("bias,gelu", [(False, False), (True, False), (True, True)])
def test_gemm_w8a8(K, m, n, k, bias, gelu):
    x = act_like(m, k, torch.bfloat16, seed=m + n + k, outliers=False)
    w = (torch.randn(n, k, generator=torch.Generator().manual_seed(k)) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=torch.Generator().manual_seed(3)) * 0.1).to(torch.bfloat16) if bias else None
    xq, xs = O.quant_block128(x)
    wq, ws = O.quant_block128(w)
    ref = O.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=gelu)
    out = K.gemm_w8a8(xq.to(DEV), xs.to(DEV), wq.to(DEV), ws.to(DEV), torch.bfloat16,
                      bias=None if b is None else b.to(DEV), gelu_tanh=gelu)
    ulp = ulp_diff_bf16(out, ref)
    if gelu:
        # the kernel evaluates tanh-GELU as x*sigmoid(2u) (exact identity, no cancellation); torch's 0.5x(1+tanh u)
        # loses its digits for x < -4 (|y| < 1e-4): there the two may differ by more than a bf16 ulp of a tiny number
        ulp = torch.where((out.float().cpu() - ref.float()).abs() <= 3e-7, torch.zeros_like(ulp), ulp)
    assert ulp.max().item() <= 1, f"max ulp {ulp.max().item()}"
    assert (ulp > 0).float().mean().item() < 0.01
    # and against the fp32 matmul of the de-quantised operands (rel-L2 <= 1e-2)
    if not gelu:
        deq = (x.float() @ w.float().t()) + (0 if b is None else b.float())
        assert rel_l2(out, deq) < 2e-2


@pytest.mark.parametrize("m,n,k", [(1000, 1536, 1536), (2050, 264, 384), (256, 256, 128), (3000, 8960, 256),
                                   (1111, 1544, 1280), (700, 512, 128), (513, 520, 2304),
                                   (9000, 2304, 256)])
@pytest.mark.parametrize("bias,gelu", [(False, False), (True, True)])
def test_gemm_variants_bit_identical(K, m, n, k, bias, gelu):
    """128x128-tile kernel, the two 256x256-tile LDS-DMA kernels (16x16x64 and 32x32x32 MFMA) and the oracle: same bits (ragged M and N tails)."""
    g = torch.Generator().manual_seed(m + n + k)
    xq = torch.randint(-128, 128, (m, k), generator=g, dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8)
    xs = torch.rand((m + 127) // 128, k // 128, generator=g) * 0.02 + 1e-3
    ws = torch.rand((n + 127) // 128, k // 128, generator=g) * 0.02 + 1e-3
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16) if bias else None
    outs = []
    for var in (1, 4, 5):
        K.set_tuning(K.TUNE_GEMM_VARIANT, var)
        try:
            outs.append(K.gemm_w8a8(xq.to(DEV), xs.to(DEV), wq.to(DEV), ws.to(DEV), torch.bfloat16,
                                    bias=None if b is None else b.to(DEV), gelu_tanh=gelu).cpu())
        finally:
            K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "GEMM kernel variants must be bit-identical"
    if m * n * k <= 1000 * 1536 * 1536 and not gelu:  # (gelu vs the oracle: test_gemm_w8a8, realistic ranges)
        ref = O.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=gelu)
        assert ulp_diff_bf16(outs[1], ref).max().item() <= 1


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 1536, 384), (3000, 8960, 256), (300, 272, 256), (1111, 1552, 1280),
                                   (77, 128, 128)])
@pytest.mark.parametrize("bias,gelu", [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize("variant", [4, 5])
def test_gemm_fused_output_quant_bit_exact(K, m, n, k, bias, gelu, variant):
    """td_gemm_w8a8_quant == td_quant_i8_block128(td_gemm_w8a8(...)): codes and scales bit for bit, incl. ragged
    M / N tails (zero-filled like the stand-alone quantiser) and realistic value ranges."""
    x = act_like(m, k, torch.bfloat16, seed=m + n + k)
    w = (torch.randn(n, k, generator=torch.Generator().manual_seed(k)) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=torch.Generator().manual_seed(3)) * 0.1).to(torch.bfloat16).to(DEV) if bias else None
    xq, xs = K.quant_i8_block128(x.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    y = K.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=gelu)
    q_ref, s_ref = K.quant_i8_block128(y)
    K.set_tuning(K.TUNE_GEMM_VARIANT, variant)
    try:
        q, s = K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=gelu)
    finally:
        K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
    assert torch.equal(s, s_ref), "scales"
    assert torch.equal(q, q_ref), f"codes differ at {(q != q_ref).sum().item()} positions"


@pytest.mark.parametrize("m,n,k,amp", [(1024, 2048, 256, 1.0), (1000, 1552, 384, 0.05), (4096, 8960, 1536, 1.0), (777, 512, 128, 30.0)])
def test_gemm_fused_gelu_table_is_bit_identical_to_inline(K, m, n, k, amp):
    """Round 5: the fused FFN GEMM looks the GELU of its (already bf16-rounded) value up in the device-built 65 536-entry table
    (csrc/gemm_w8a8_fi.hip: g_gelu_tab_bf16) instead of evaluating it; TD_TUNE_GELU_TABLE = 1 keeps the inline form.  Same
    codes and scales bit for bit over tiny (|x| ~ 1e-3: gelu(x) = x / 2), ordinary and saturating (|x| ~ 1e2: x or -0) ranges."""
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) * amp).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=g) * 0.3 * amp).to(torch.bfloat16).to(DEV)
    xq, xs = K.quant_i8_block128(x.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    K.set_tuning(K.TUNE_GELU_TABLE, 1)
    try:
        q1, s1 = K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
    finally:
        K.set_tuning(K.TUNE_GELU_TABLE, 0)
    q0, s0 = K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
    assert torch.equal(s0, s1), "scales"
    assert torch.equal(q0, q1), f"codes differ at {(q0 != q1).sum().item()} positions"


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 1536, 384), (300, 264, 256), (1111, 1544, 1280), (4096, 1536, 1536)])
@pytest.mark.parametrize("bias,gated", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("variant", [4, 5])
def test_gemm_fused_residual_bit_exact(K, m, n, k, bias, gated, variant):
    """td_gemm_w8a8_residual == td_gated_residual(x, td_gemm_w8a8(...), gate) bit for bit (ragged M / N tails)."""
    g = torch.Generator().manual_seed(m + n)
    a = act_like(m, k, torch.bfloat16, seed=m + n + k)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).to(DEV) if bias else None
    gate = (torch.randn(1, n, generator=g) * 0.5).to(DEV) if gated else None
    x0 = torch.randn(m, n, generator=g).to(torch.bfloat16).to(DEV)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    ref = K.gated_residual_(x0.clone(), K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), gate)
    K.set_tuning(K.TUNE_GEMM_VARIANT, variant)
    try:
        out = K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate)
    finally:
        K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
    assert torch.equal(out, ref), f"differ at {(out != ref).sum().item()} positions"


@pytest.mark.parametrize("m,n,k", [(1000, 1536, 1536), (2050, 264, 384), (1111, 1544, 1280), (3000, 512, 8960)])
@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("G", [2, 4, 8])
@pytest.mark.parametrize("outliers", [False, True])
def test_gemm_fast_dequant_within_stated_bound(K, m, n, k, variant, G, outliers):
    """TD_TUNE_GEMM_FAST = G (one-VALU dequant, re-centred every G K blocks; opt-in, never the default) against the
    exact kernel: the fp32 accumulators differ by at most 0.75 (G+1) sum_k s_k (csrc/gemm_w8a8_fi.hip), i.e. the 16-bit
    outputs by that plus one rounding; on realistic operands the two agree to rel-L2 < 1.5e-3 (measured 3e-4..8e-4: ulp
    flips of the bf16 result).  Zero bias: the reference re-rounds after the bias add (ops/core.py:408-412), which can
    amplify a one-ulp flip of the intermediate without bound when bias ~ -acc; that is a property of the two-rounding
    epilogue, not of the accumulation under test."""
    x = act_like(m, k, torch.bfloat16, seed=m + n + k, outliers=outliers).to(DEV)
    w = (torch.randn(n, k, generator=torch.Generator().manual_seed(k)) / k ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    xq, xs = K.quant_i8_block128(x)
    wq, ws = K.quant_i8_block128(w)
    K.set_tuning(K.TUNE_GEMM_VARIANT, variant)
    try:
        K.set_tuning(K.TUNE_GEMM_FAST, 1)
        exact = K.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b).float()
        K.set_tuning(K.TUNE_GEMM_FAST, G)
        fast = K.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b).float()
    finally:
        K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
        K.set_tuning(K.TUNE_GEMM_FAST, 0)
    ssum = (xs[:, None, :] * ws[None, :, :]).sum(-1)
    bound = 0.75 * (G + 1) * ssum.repeat_interleave(128, 0)[:m].repeat_interleave(128, 1)[:, :n]
    one_ulp = exact.abs().clamp_min(1e-30) * 2.0 ** -7      # a bf16 ulp is <= 2^-7 |x|
    d = (fast - exact).abs()
    assert (d <= bound + one_ulp).all(), f"bound exceeded by {(d / (bound + one_ulp)).max().item():.2f}x"
    assert rel_l2(fast, exact) < 1.5e-3


def test_gemm_rejects_bad_k(K):
    from turbodiffusion_amd._lib import TurboDiffusionAMDError
    a = torch.zeros(128, 192, dtype=torch.int8, device=DEV)
    s = torch.ones(1, 1, device=DEV)
    with pytest.raises((TurboDiffusionAMDError, AssertionError)):
        K.gemm_w8a8(a, s, a, s)


@pytest.mark.parametrize("m,n", [(7, 1536), (300, 256), (64, 5120), (33, 1024), (5, 8)])
def test_rmsnorm(K, m, n):
    x = act_like(m, n, torch.bfloat16, seed=n)
    w = torch.rand(n, generator=torch.Generator().manual_seed(1)) + 0.5
    ref = O.rmsnorm_fast(x, w, 1e-6)
    out = K.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    assert ulp_diff_bf16(out, ref).max().item() <= 1
    # fp32 in / fp32 out (the reference's API-level contract, ops/core.py:139)
    out32 = K.rmsnorm(x.float().to(DEV), w.to(DEV), 1e-6)
    ref32 = O.rmsnorm_fast(x.float(), w, 1e-6)
    torch.testing.assert_close(out32.cpu(), ref32, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("m,n", [(7, 1536), (300, 256), (64, 5120)])
@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("mod", [False, True])
@pytest.mark.parametrize("triton_variance", [True, False])
def test_layernorm_modulate(K, m, n, affine, mod, triton_variance):
    """triton_variance: the reference's Triton LayerNorm (variance over next_power_of_2(n) columns, the phantom ones
    contributing mean^2 — ops/core.py:213-224) vs the textbook one; rows carry an offset so that the two differ."""
    g = torch.Generator().manual_seed(n + m)
    x = (act_like(m, n, torch.bfloat16, seed=n + 1).float() + 0.375).bfloat16()
    w = (torch.rand(n, generator=g) + 0.5) if affine else None
    b = (torch.randn(n, generator=g) * 0.1) if affine else None
    ref = O.layernorm_fast(x, w, b, 1e-6, triton_variance=triton_variance)
    pad = K.triton_ln_pad_cols(n) if triton_variance else 0
    assert pad == {1536: 512, 256: 0, 5120: 3072}[n] * int(triton_variance)
    scale = shift = None
    if mod:
        scale = torch.randn(1, n, generator=g) * 0.3
        shift = torch.randn(1, n, generator=g) * 0.3
        ref = O.modulate(ref, scale, shift)
    out = K.layernorm(x.to(DEV), None if w is None else w.to(DEV), None if b is None else b.to(DEV), 1e-6,
                      None if scale is None else scale.to(DEV), None if shift is None else shift.to(DEV), pad_cols=pad)
    ulp = ulp_diff_bf16(out, ref)
    # modulate amplifies a 1-ulp difference of the rounded norm output by (1+scale): allow 2 there
    assert ulp.max().item() <= (2 if mod else 1), f"max ulp {ulp.max().item()}"
    assert (ulp > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("m,n", [(7, 1536), (300, 256), (1000, 1536), (129, 1024), (256, 520), (4096, 1536), (300, 5120),
                                 (1002, 264)])
@pytest.mark.parametrize("affine,mod", [(False, True), (True, False), (False, False), (True, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_quant_bit_exact(K, m, n, affine, mod, dtype):
    """td_layernorm_quant == td_quant_i8_block128(td_layernorm(...)): INT8 codes and fp32 scales bit for bit
    (ragged row tails, two batch entries with different AdaLN vectors, outlier channels)."""
    x = act_like(m, n, dtype, seed=m + n).to(DEV)
    g = torch.Generator().manual_seed(n)
    w = (1 + 0.1 * torch.randn(n, generator=g)).to(DEV) if affine else None
    b = (0.05 * torch.randn(n, generator=g)).to(DEV) if affine else None
    nb = 2 if (mod and m % 2 == 0) else 1
    sc = (0.2 * torch.randn(nb, n, generator=g)).to(DEV) if mod else None
    sh = (0.2 * torch.randn(nb, n, generator=g)).to(DEV) if mod else None
    pad = K.triton_ln_pad_cols(n)      # (0 for the power-of-two widths)
    y = K.layernorm(x, w, b, 1e-6, sc, sh, rows_per_batch=m // nb if mod else 0, pad_cols=pad)
    q_ref, s_ref = K.quant_i8_block128(y)
    q, s = K.layernorm_quant(x, w, b, 1e-6, sc, sh, rows_per_batch=m // nb if mod else 0, pad_cols=pad)
    assert torch.equal(s, s_ref), "scales"
    assert torch.equal(q, q_ref), f"codes differ at {(q != q_ref).sum().item()} positions"


def test_gated_residual_bit_exact(K):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 1536, generator=g).bfloat16()
    y = torch.randn(300, 1536, generator=g).bfloat16()
    gate = torch.randn(1, 1536, generator=g)
    ref = O.gated_residual(x, y, gate)
    out = K.gated_residual_(x.clone().to(DEV), y.to(DEV), gate.to(DEV))
    assert torch.equal(out.cpu(), ref)
    ref2 = x + y
    out2 = K.gated_residual_(x.clone().to(DEV), y.to(DEV), None)
    assert torch.equal(out2.cpu(), ref2)


@pytest.mark.parametrize("L,H", [(100, 2), (333, 12)])
def test_qk_norm_rope(K, L, H):
    D = 128
    g = torch.Generator().manual_seed(L)
    src = torch.randn(L, 3 * H * D, generator=g).bfloat16()
    w = torch.rand(H * D, generator=g) + 0.5
    freqs = O.rope_freqs(5, 9, 11, D)[:L] if L <= 495 else None
    qn = O.rmsnorm_fast(src[:, : H * D], w, 1e-6)
    ref = O.rope_apply(qn.view(1, L, H, D), freqs)[0].transpose(0, 1)  # [H, L, D]
    cos, sin = torch.cos(freqs).float(), torch.sin(freqs).float()
    out = K.qk_norm_rope(src.to(DEV), 0, H, D, w.to(DEV), cos.to(DEV), sin.to(DEV), 1e-6)
    ulp = ulp_diff_bf16(out, ref.contiguous())
    assert ulp.max().item() <= 2 and (ulp > 0).float().mean().item() < 0.02
    # plain relayout of the V columns
    v = K.qk_norm_rope(src.to(DEV), 2 * H * D, H, D, None, None, None, 1e-6)
    assert torch.equal(v.cpu(), src[:, 2 * H * D:].view(L, H, D).transpose(0, 1))


# ---------------------------------------------------------------- seam 1: the pybind-signature stand-in
def test_turbo_diffusion_ops_compat_drives_the_reference_call_sequence(K):
    """turbodiffusion_amd.compat.turbo_diffusion_ops exports quant_cuda / gemm_cuda / rms_norm_cuda / layer_norm_cuda with
    the pybind signatures (ops/bindings.cpp:6-16).  The reference's int8_linear call sequence (ops/core.py:49-57:
    torch.zeros y, quant_cuda(x, None, None), gemm_cuda(x_q, x_s, w_q, w_s, y)) through it == the oracle, as are the
    norms' optional-output conventions."""
    import sys
    from turbodiffusion_amd.compat import turbo_diffusion_ops as T
    sys.modules.setdefault("turbo_diffusion_ops", T)
    from turbo_diffusion_ops import gemm_cuda, layer_norm_cuda, quant_cuda, rms_norm_cuda   # as ops/core.py:9 does
    x = act_like(300, 512, torch.bfloat16, seed=1).to(DEV)
    w = (torch.randn(264, 512, generator=torch.Generator().manual_seed(2)) / 22).bfloat16()
    w_q, w_s = O.quant_block128(w)
    y = torch.zeros(300, 264, dtype=x.dtype, device=DEV)
    x_q, x_s = quant_cuda(x, None, None)
    assert gemm_cuda(x_q, x_s, w_q.to(DEV), w_s.to(DEV), y) is None
    xq_ref, xs_ref = O.quant_block128(x.cpu())
    assert torch.equal(x_q.cpu(), xq_ref) and torch.equal(x_s.cpu(), xs_ref)
    assert ulp_diff_bf16(y, O.gemm_w8a8(xq_ref, xs_ref, w_q, w_s)).max().item() <= 1
    # TurboT2AV's newer entry points (LTX-2 ltx_distillation/acceleration.py:753-769): the raster hints are accepted and ignored,
    # the bias variant is Int8Linear's cast / + bias / cast
    from turbo_diffusion_ops import gemm_cuda_swizzle, gemm_cuda_swizzle_bias
    y2 = torch.zeros_like(y)
    assert gemm_cuda_swizzle(x_q, x_s, w_q.to(DEV), w_s.to(DEV), y2, 1, 3) is None and torch.equal(y2, y)
    bias = (torch.randn(264, generator=torch.Generator().manual_seed(9)) * 0.1).bfloat16().to(DEV)
    y3 = torch.zeros_like(y)
    gemm_cuda_swizzle_bias(x_q, x_s, w_q.to(DEV), w_s.to(DEV), y3, bias, 0, 0)
    assert torch.equal(y3, (y.float() + bias.float()).bfloat16())
    # caller-provided outputs are written in place and returned
    oq, os_ = torch.empty_like(x_q), torch.empty_like(x_s)
    r = quant_cuda(x, oq, os_)
    assert r[0] is oq and r[1] is os_ and torch.equal(oq, x_q) and torch.equal(os_, x_s)
    with pytest.raises(RuntimeError):
        quant_cuda(x.float(), None, None)
    with pytest.raises(RuntimeError):
        gemm_cuda(x_q, x_s, w_q.to(DEV), w_s.to(DEV), y.float())
    # norms: fp32 in / fp32 out, optional weight / bias / output
    xf = torch.randn(33, 1536, generator=torch.Generator().manual_seed(3)).to(DEV)
    wn = (torch.rand(1536, generator=torch.Generator().manual_seed(4)) + 0.5).to(DEV)
    bn = (torch.randn(1536, generator=torch.Generator().manual_seed(5)) * 0.1).to(DEV)
    torch.testing.assert_close(rms_norm_cuda(xf, 1e-6, wn, None).cpu(), O.rmsnorm_fast(xf.cpu(), wn.cpu(), 1e-6), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(rms_norm_cuda(xf, 1e-6, None, None).cpu(), O.rmsnorm_fast(xf.cpu(), torch.ones(1536), 1e-6), rtol=2e-6, atol=1e-6)
    out = torch.empty_like(xf)
    assert layer_norm_cuda(xf, 1e-6, wn, bn, out) is out
    torch.testing.assert_close(out.cpu(), O.layernorm_fast(xf.cpu(), wn.cpu(), bn.cpu(), 1e-6, triton_variance=False), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(layer_norm_cuda(xf, 1e-6, None, None, None).cpu(), O.layernorm_fast(xf.cpu(), None, None, 1e-6, triton_variance=False), rtol=2e-5, atol=2e-6)


# ---------------------------------------------------------------- row statistics from the GEMM epilogue
@pytest.mark.parametrize("m,n,k", [(1300, 1536, 384), (2100, 5120, 256), (1024, 320, 128)])
@pytest.mark.parametrize("residual", [True, False])
def test_gemm_row_stats_epilogue(K, m, n, k, residual):
    """td_gemm_w8a8_stats: the output bits are those of td_gemm_w8a8 / td_gemm_w8a8_residual, and the per-piece (mean, M2)
    merged by td_row_stats_finalize give the LayerNorm statistics of the stored rows (vs an fp64 evaluation
    of the stored 16-bit values) and the RMSNorm statistic td_rms_stats computes."""
    g = torch.Generator().manual_seed(m + n)
    a = act_like(m, k, torch.bfloat16, seed=m + n + k)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    gate = (torch.randn(1, n, generator=g) * 0.5).to(DEV)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    if residual:
        x0 = (torch.randn(m, n, generator=g) * 2 + 0.3).to(torch.bfloat16).to(DEV)
        ref = K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate)
        out, part = K.gemm_w8a8_stats(aq, as_, wq, ws, b, x=x0.clone(), gate=gate)
    else:
        ref = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
        out, part = K.gemm_w8a8_stats(aq, as_, wq, ws, b)
    assert torch.equal(out, ref)
    st = K.row_stats_finalize(part, n, 1e-6)
    r64 = ref.double()
    mean, var = r64.mean(-1), r64.var(-1, unbiased=False)
    torch.testing.assert_close(st[:, 0].double(), mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st[:, 1].double(), 1.0 / torch.sqrt(var + 1e-6), rtol=2e-5, atol=0)
    rstd = K.row_stats_finalize(part, n, 1e-6, rms=True)
    torch.testing.assert_close(rstd, K.rms_stats(ref, n, 1e-6), rtol=2e-6, atol=0)
    # the reference's Triton variance (pad_cols phantom columns contributing mean^2 each, K.triton_ln_pad_cols)
    pad = K.triton_ln_pad_cols(n)
    stp = K.row_stats_finalize(part, n, 1e-6, pad_cols=pad)
    torch.testing.assert_close(stp[:, 1].double(), 1.0 / torch.sqrt(var + pad * mean * mean / n + 1e-6), rtol=2e-5, atol=0)
    if n <= K.LNQ_MAX_N:
        qa, sa = K.layernorm_quant(ref, None, None, 1e-6, pad_cols=pad)
        qb, sb = K.layernorm_quant(ref, None, None, 1e-6, stats=stp)
        assert (qa != qb).float().mean().item() < 2e-3 and (qa.int() - qb.int()).abs().max().item() <= 1
    # LayerNorm -> INT8 with the supplied statistics against its own statistics pass: same scales / codes except where a
    # last-bit difference of (mean, rstd) moves a value across a rounding boundary
    sc, sh = (0.2 * torch.randn(1, n, generator=g)).to(DEV), (0.2 * torch.randn(1, n, generator=g)).to(DEV)
    if n <= K.LNQ_MAX_N:
        q0, s0 = K.layernorm_quant(ref, None, None, 1e-6, sc, sh, rows_per_batch=m)
        q1, s1 = K.layernorm_quant(ref, None, None, 1e-6, sc, sh, rows_per_batch=m, stats=st)
        assert (q0 != q1).float().mean().item() < 2e-3 and (q0.int() - q1.int()).abs().max().item() <= 1
        assert (s0 != s1).float().mean().item() < 0.05
        torch.testing.assert_close(s0, s1, rtol=1e-2, atol=0)


@pytest.mark.parametrize("offset,spread", [(100.0, 1.0), (-300.0, 4.0), (1000.0, 8.0)])
@pytest.mark.parametrize("residual", [True, False])
def test_row_stats_keep_their_digits_on_dc_heavy_rows(K, offset, spread, residual):
    """Rows whose mean is 17-1000x their spread (what a DC-heavy channel pattern does to a residual stream): a one-pass
    E[x^2] - mean^2 in fp32 loses 4-6 of its 7 digits there (mean^2 / var = 1e4 ... 1e6) — the STATS epilogue carries
    per-piece (mean, M2) of shifted values and td_row_stats_finalize merges them Chan-style, so (mean, rstd) must agree
    with an fp64 evaluation of the stored 16-bit values as tightly as for zero-mean rows (same rtol as
    test_gemm_row_stats_epilogue), like the reference's two-pass statistics (ops/core.py:293-335)."""
    m, n, k = 1100, 1536, 256
    g = torch.Generator().manual_seed(int(abs(offset)))
    a = act_like(m, k, torch.bfloat16, seed=77)
    w = (torch.randn(n, k, generator=g) / k ** 0.5 * spread).to(torch.bfloat16)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    if residual:     # the offset sits in the residual stream, the GEMM adds the spread
        b = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        x0 = (offset + spread * torch.randn(m, n, generator=g)).to(torch.bfloat16).to(DEV)
        out, part = K.gemm_w8a8_stats(aq, as_, wq, ws, b, x=x0, gate=torch.ones(1, n, device=DEV))
    else:            # the offset comes in through the bias
        b = torch.full((n,), offset, dtype=torch.bfloat16, device=DEV)
        out, part = K.gemm_w8a8_stats(aq, as_, wq, ws, b)
    r64 = out.double()
    mean, var = r64.mean(-1), r64.var(-1, unbiased=False)
    # |mean| / sigma >= 10 on every row (17 ... 1000 here): E[x^2] - mean^2 would lose >= 2 of fp32's 7 digits
    assert (mean.abs() / var.sqrt()).min().item() > 10, "the rows are not DC-heavy: the test does not exercise the hazard"
    st = K.row_stats_finalize(part, n, 1e-6)
    torch.testing.assert_close(st[:, 0].double(), mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st[:, 1].double(), 1.0 / torch.sqrt(var + 1e-6), rtol=2e-5, atol=0)
    # ... and they are the two-pass kernel's statistics to its own rounding: the INT8 codes downstream agree
    sc, sh = (0.2 * torch.randn(1, n, generator=g)).to(DEV), (0.2 * torch.randn(1, n, generator=g)).to(DEV)
    q0, s0 = K.layernorm_quant(out, None, None, 1e-6, sc, sh, rows_per_batch=m)
    q1, s1 = K.layernorm_quant(out, None, None, 1e-6, sc, sh, rows_per_batch=m, stats=st)
    assert (q0 != q1).float().mean().item() < 2e-3 and (q0.int() - q1.int()).abs().max().item() <= 1
    rstd = K.row_stats_finalize(part, n, 1e-6, rms=True)
    torch.testing.assert_close(rstd, K.rms_stats(out, n, 1e-6), rtol=2e-6, atol=0)



# ---------------------------------------------------------------- V^T tiles from the q|k|v GEMM's epilogue
@pytest.mark.parametrize("m,dim,k", [(1000, 256, 256), (2100, 512, 384), (1024, 1536, 128)])
@pytest.mark.parametrize("dt,vt_dt", [(torch.bfloat16, torch.float16), (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16)])
def test_gemm_vt_epilogue_is_gemm_plus_v_transpose(K, m, dim, k, dt, vt_dt):
    """td_gemm_w8a8_vt: q|k columns = td_gemm_w8a8's bits; the V columns leave as exactly the tiles td_v_transpose makes
    of td_gemm_w8a8's V columns (incl. the fp16 cast, the bit-2/3 key permutation, zero tail keys, a ragged last 64-key
    block and a partial last 256-row tile)."""
    g = torch.Generator().manual_seed(m + dim)
    a = act_like(m, k, torch.bfloat16, seed=m + k)
    w = (torch.randn(3 * dim, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(3 * dim, generator=g) * 0.1).to(dt).to(DEV)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    ref = K.gemm_w8a8(aq, as_, wq, ws, dt, bias=b)
    H = dim // 128
    vt_ref = K.v_transpose(ref[:, 2 * dim:], 128, 3 * dim, m, H, 128, vt_dt)
    d, vt = K.gemm_w8a8_vt(aq, as_, wq, ws, b, 2 * dim, vt_dt, out_dtype=dt)
    assert torch.equal(d[:, :2 * dim], ref[:, :2 * dim])
    assert vt.shape == vt_ref.shape and vt.dtype == vt_dt
    assert torch.equal(vt.view(torch.int16), vt_ref.view(torch.int16))


# ---------------------------------------------------------------- round 5: the 128-row form of the LDS-DMA kernel (NI = 4)
@pytest.mark.parametrize("m,n,k", [(4096, 1536, 1536), (4000, 1544, 384), (2100, 512, 256), (4096, 1536, 8960), (1100, 4608, 1536),
                                   (4096, 8960, 1536), (300, 272, 256), (129, 768, 128), (77, 64, 640)])
def test_gemm_128_row_tile_form_is_bit_identical(K, m, n, k):
    """TD_TUNE_GEMM_VARIANT = 6 forces the 128 x 256 tile (csrc/gemm_w8a8_fi.hip, NI = 4: the eight waves on 64 x 64 each) — the
    form the per-rank GEMMs of a sequence shard take automatically: every epilogue (plain + GELU, gated / plain residual,
    row-statistics partials with and without residual, fused quantiser with the GELU as a table and inline, V^T tiles) gives
    the bits of the 256 x 256 tile (variant 4), ragged M / N tails included — partials too: a wave's 64-column piece is summed
    in the same order in both forms."""
    g = torch.Generator().manual_seed(m + n + k)
    a = act_like(m, k, torch.bfloat16, seed=m + k)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    gate = (torch.randn(1, n, generator=g) * 0.5).to(DEV)
    x0 = (torch.randn(m, n, generator=g) * 2 + 0.3).to(torch.bfloat16).to(DEV)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    outs = {}
    for variant in (4, 6, 8, 0, 104, 108):      # 0: the automatic plan — for (4096, 8960, 1536) the MIXED one (two rounds of 256-row tiles + 128-row tiles)
        # 8 (round 6): the FOUR-wave form (128 x 256 tile, two workgroups per CU); 104 / 108: forms 4 / 8 with the one-VALU dequant
        K.set_tuning(K.TUNE_GEMM_VARIANT, variant % 100)
        K.set_tuning(K.TUNE_GEMM_FAST, 8 if variant >= 100 else 1)     # (8: the period whose instantiations cover every epilogue and form)
        try:
            r = {"plain": K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), "gelu": K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True),
                 "nobias": K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16),
                 "res": K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate),
                 "res_nogate": K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=None)}
            if n % 64 == 0:
                r["stats_x"], r["part_x"] = K.gemm_w8a8_stats(aq, as_, wq, ws, b, x=x0.clone(), gate=gate)
                r["stats_y"], r["part_y"] = K.gemm_w8a8_stats(aq, as_, wq, ws, b)
            if n % 16 == 0:
                r["q"], r["qs"] = K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
                r["q0"], r["qs0"] = K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b)
                K.set_tuning(K.TUNE_GELU_TABLE, 1)
                r["qi"], r["qsi"] = K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
                K.set_tuning(K.TUNE_GELU_TABLE, 0)
            if n % 768 == 0:      # q | k | v of n / 3 columns each, heads of 128: V leaves as the attention kernel's tiles
                r["vt_d"], r["vt"] = K.gemm_w8a8_vt(aq, as_, wq, ws, b, 2 * n // 3, torch.float16)
                r["vt_d"] = r["vt_d"][:, :2 * n // 3].clone()
        finally:
            K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
            K.set_tuning(K.TUNE_GELU_TABLE, 0)
            K.set_tuning(K.TUNE_GEMM_FAST, 0)
        outs[variant] = r
    for base, other in ((4, 6), (4, 8), (4, 0), (104, 108)):
        for key in outs[base]:
            if other == 0 and key.startswith("part_"):
                continue      # (a problem this small goes to the 128x128 kernel automatically: its partials agree to rounding, not bit for bit)
            assert torch.equal(outs[base][key].view(torch.uint8) if outs[base][key].dtype == torch.float16 else outs[base][key],
                               outs[other][key].view(torch.uint8) if outs[other][key].dtype == torch.float16 else outs[other][key]), (other, key)
    if "q" in outs[6]:
        assert torch.equal(outs[6]["q"], outs[6]["qi"]) and torch.equal(outs[6]["qs"], outs[6]["qsi"])
    # the one-VALU dequant of EVERY epilogue (round 6) stays within a bf16 rounding step of the exact form
    for key in ("plain", "gelu", "res", "res_nogate"):
        e, f = outs[4][key].float(), outs[104][key].float()
        assert ((e - f).norm() / e.norm()).item() < 2e-3, key


# ---------------------------------------------------------------- small problems: the fused epilogues on the 128x128 kernel
@pytest.mark.parametrize("m,n,k", [(4096, 1536, 1536), (4000, 1536, 384), (2100, 512, 256), (4096, 1536, 8960)])
def test_small_problem_gemm_epilogues_match_the_256_tile_kernel(K, m, n, k):
    """Problems that leave more than half of the CUs without a 256x256 tile (the per-rank shapes of an 8-way sequence split:
    M = 4096, N = 1536 -> 96 tiles) run the fused epilogues on the 128x128 kernel (csrc/gemm_w8a8.hip: G_RES, G_STATS, G_QOUT).
    Residual and quantiser epilogues: bit-identical to the 256x256 kernel's (TD_TUNE_GEMM_VARIANT = 4 forces that one);
    row statistics: the same (mean, rstd) to rounding."""
    g = torch.Generator().manual_seed(m + n + k)
    a = act_like(m, k, torch.bfloat16, seed=m + k)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    gate = (torch.randn(1, n, generator=g) * 0.5).to(DEV)
    x0 = (torch.randn(m, n, generator=g) * 2 + 0.3).to(torch.bfloat16).to(DEV)
    aq, as_ = K.quant_i8_block128(a.to(DEV))
    wq, ws = K.quant_i8_block128(w.to(DEV))
    outs = {}
    for variant in (4, 1):          # 4: the 256x256 kernel; 1: the 128x128 kernel (the automatic choice for these shapes until round 5)
        K.set_tuning(K.TUNE_GEMM_VARIANT, variant)
        try:
            r = {"plain": K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True),
                 "res": K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate),
                 "res_nogate": K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=None)}
            r["stats_x"], part = K.gemm_w8a8_stats(aq, as_, wq, ws, b, x=x0.clone(), gate=gate)
            r["stats"] = K.row_stats_finalize(part, n, 1e-6, pad_cols=K.triton_ln_pad_cols(n))
            r["stats_y"], part2 = K.gemm_w8a8_stats(aq, as_, wq, ws, b)
            r["rms"] = K.row_stats_finalize(part2, n, 1e-6, rms=True)
            if n % 128 == 0:
                r["q"], r["qs"] = K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
        finally:
            K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
        outs[variant] = r
    big, small = outs[4], outs[1]
    for key in ("plain", "res", "res_nogate", "stats_x", "stats_y") + (("q", "qs") if n % 128 == 0 else ()):
        assert torch.equal(big[key], small[key]), key
    torch.testing.assert_close(small["stats"], big["stats"], rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(small["rms"], big["rms"], rtol=2e-6, atol=0)
    r64 = small["stats_x"].double()
    torch.testing.assert_close(small["stats"][:, 0].double(), r64.mean(-1), rtol=1e-5, atol=1e-6)
