"""MFMA projection kernel (C ABI segger_linear_fwd) against an fp32 torch reference of the same op
on the same bf16/f16-rounded inputs.  Tolerance: fp32 accumulation of K<=384 products, one rounding
of the result to the storage dtype -> |d| <= 2^-8 * |ref| + 1e-3 (bf16), 2^-10 (f16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,m", [(64, 64), (128, 64), (128, 384), (256, 384), (384, 256), (384, 128), (256, 64)])
@pytest.mark.parametrize("n", [1, 127, 128, 1000, 4133])
def test_linear_forward(cuda, dtype, k, m, n):
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(n + k + m)
    x = torch.randn(n, k, device=cuda, generator=g).to(dtype)
    w = (torch.randn(m, k, device=cuda, generator=g) / k ** 0.5)
    b = torch.randn(m, device=cuda, generator=g)
    assert ops.linear_supported(k, m, dtype)
    y = ops.linear(x, w, b)
    ref = x.float() @ w.to(dtype).float().t() + b
    rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert y.dtype == dtype and y.shape == (n, m)
    assert ((y.float() - ref).abs() <= rel * ref.abs() + 1e-3).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,m", [(128, 384), (128, 128), (128, 64), (64, 384), (64, 64)])
@pytest.mark.parametrize("n", [262_144, 300_007])          # >= 2 tiles per wave of the persistent grid; 300007: partial last tile
def test_linear_forward_resident_weights(cuda, dtype, k, m, n):
    """Large row counts take the persistent resident-W kernel (linear_res_kernel): same arithmetic as the chunked kernel
    -- checked bit for bit against small-n launches over slices, which take the chunked one -- with and without
    bias, on a strided input view, rows behind the output intact."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(n + k + m)
    big = torch.randn(n, k + 64, device=cuda, generator=g).to(dtype)
    x = big[:, 64:]                                        # row stride k + 64
    w = (torch.randn(m, k, device=cuda, generator=g) / k ** 0.5).to(dtype)
    b = torch.randn(m, device=cuda, generator=g)
    guard = torch.full((n + 64, m), 7.0, device=cuda, dtype=dtype)        # rows behind the output must stay untouched
    y = ops.linear_fwd_launch(x, w, b, out=guard[:n])
    assert torch.equal(guard[n:], torch.full((64, m), 7.0, device=cuda, dtype=dtype))
    ref = x.float() @ w.float().t() + b
    rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((y.float() - ref).abs() <= rel * ref.abs() + 1e-3).all()
    for lo in (0, 131_072 + 17, n - 5000):                 # chunked kernel on slices (n < the persistent threshold)
        sl = slice(lo, lo + 5000)
        assert torch.equal(ops.linear_fwd_launch(x[sl], w, b), y[sl])
    y0 = ops.linear_fwd_launch(x, w, None)
    assert torch.equal(ops.linear_fwd_launch(x[-4133:], w, None), y0[-4133:])
    # a column window of a wider matrix as the output (row stride m + 128): the columns beside it stay untouched
    wide = torch.full((n, m + 128), 3.0, device=cuda, dtype=dtype)
    ops.linear_fwd_launch(x, w, b, out=wide[:, 64:64 + m])
    assert torch.equal(wide[:, 64:64 + m], y) and bool((wide[:, :64] == 3.0).all()) and bool((wide[:, 64 + m:] == 3.0).all())


@pytest.mark.parametrize("k,m", [(64, 64), (128, 64), (128, 384), (256, 384), (384, 256), (384, 128), (256, 64), (64, 128)])
@pytest.mark.parametrize("n", [1, 127, 128, 1000, 4133])
def test_linear_forward_fp32(cuda, k, m, n):
    """fp32 storage on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation) against float64: the error of an
    fp32 dot product of K <= 384 terms, <= 1e-6 * sum |x||w| (K * 2^-24 = 2.3e-5 is the worst case; sums of random signs
    stay far below it), and the result must agree with torch's own fp32 GEMM to the same bound."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(n + k + m)
    x = torch.randn(n, k, device=cuda, generator=g)
    w = torch.randn(m, k, device=cuda, generator=g) / k ** 0.5
    b = torch.randn(m, device=cuda, generator=g)
    assert ops.linear_supported(k, m, torch.float32)
    y = ops.linear(x, w, b)
    assert y.dtype == torch.float32 and y.shape == (n, m)
    ref = x.double() @ w.double().t() + b.double()
    bound = 2e-6 * (x.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-7
    assert bool(((y.double() - ref).abs() <= bound).all())
    big = torch.randn(n, 3 * k, device=cuda, generator=g)          # a column window of a wider matrix
    yv = ops.linear(big[:, k:2 * k], w, None)
    assert torch.allclose(yv.double(), big[:, k:2 * k].double() @ w.double().t(), rtol=0, atol=float(bound.max()))


@pytest.mark.parametrize("m,k", [(384, 256), (384, 128), (384, 64), (192, 64), (192, 256), (128, 256), (128, 128),
                                 (128, 64), (64, 256), (64, 128), (64, 64)])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 1000, 70_001])
def test_linear_wgrad_fp32_matches_fp64(cuda, m, k, n):
    """fp32 weight / bias gradients on the exact-fp32 MFMA against float64 (bound as for the 16-bit kernel: 1e-5 of
    sum |dY||X|), exact on small integers, deterministic."""
    from segger_amd import ops
    assert ops.linear_wgrad_supported(m, k, torch.float32)
    g = torch.Generator(device=cuda).manual_seed(n + m + k)
    gy = torch.randn(n, m, device=cuda, generator=g) + 0.1
    x = torch.randn(n, k, device=cuda, generator=g)
    gw, gb = ops.linear_wgrad_launch(gy, x)
    assert gw.shape == (m, k) and gb.shape == (m,)
    ref_w = gy.double().t() @ x.double()
    assert bool(((gw.double() - ref_w).abs() <= 1e-5 * (gy.double().abs().t() @ x.double().abs()) + 1e-6).all())
    assert bool(((gb.double() - gy.double().sum(0)).abs() <= 1e-5 * gy.double().abs().sum(0) + 1e-6).all())
    gw2, gb2 = ops.linear_wgrad_launch(gy, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    r = torch.arange(n, device=cuda)
    iy = ((r[:, None] * 3 + torch.arange(m, device=cuda)[None] * 5) % 7 - 3).float()
    ix = ((r[:, None] * 2 + torch.arange(k, device=cuda)[None] * 11) % 5 - 2).float()
    if n <= 1000:
        ew, eb = ops.linear_wgrad_launch(iy, ix)
        assert torch.equal(ew, iy.t() @ ix) and torch.equal(eb, iy.sum(0))


@pytest.mark.parametrize("k,m", [(256, 384), (128, 384), (128, 64)])
def test_linear_autograd_fp32(cuda, k, m):
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(k)
    n = 40001
    x = torch.randn(n, k, device=cuda, generator=g).requires_grad_(True)
    w = (torch.randn(m, k, device=cuda, generator=g) / k ** 0.5).requires_grad_(True)
    b = torch.randn(m, device=cuda, generator=g).requires_grad_(True)
    gy = torch.randn(n, m, device=cuda, generator=g)
    ops.linear(x, w, b).backward(gy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    (xr @ wr.t() + br).backward(gy.double())
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-5, atol=1e-5)
    assert (w.grad.double() - wr.grad).abs().max().item() < 1e-5 * (gy.double().abs().t() @ x.detach().double().abs()).max().item()
    assert torch.allclose(b.grad.double(), br.grad, rtol=1e-5, atol=1e-3)


def test_linear_strided_input_and_leading_dims(cuda):
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(0)
    big = torch.randn(777, 3 * 128, device=cuda, generator=g).to(torch.bfloat16)
    x = big[:, 128:256]                                   # row stride 384, 16-byte aligned offset
    w = torch.randn(64, 128, device=cuda, generator=g) / 11
    y = ops.linear(x, w, None)
    ref = x.float() @ w.to(torch.bfloat16).float().t()
    assert torch.allclose(y.float(), ref, rtol=2 ** -8, atol=1e-3)
    x3 = torch.randn(50, 2, 256, device=cuda, generator=g).to(torch.bfloat16)
    w3 = torch.randn(64, 256, device=cuda, generator=g) / 16
    y3 = ops.linear(x3, w3, None)
    assert y3.shape == (50, 2, 64)
    assert torch.allclose(y3.float(), x3.float() @ w3.to(torch.bfloat16).float().t(), rtol=2 ** -8, atol=1e-3)


@pytest.mark.parametrize("n", [3000, 40001])          # 40001 rows: batched (split-reduction) weight gradient + remainder
@pytest.mark.parametrize("k,m", [(256, 384), (128, 384), (128, 64), (256, 128)])
def test_linear_autograd(cuda, k, m, n):
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(k)
    x = torch.randn(n, k, device=cuda, generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(m, k, device=cuda, generator=g) / k ** 0.5).requires_grad_(True)
    b = torch.randn(m, device=cuda, generator=g).requires_grad_(True)
    gy = torch.randn(n, m, device=cuda, generator=g).to(torch.bfloat16)
    y = ops.linear(x, w, b)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    (xr @ wr.t() + br).backward(gy.float())
    assert torch.allclose(x.grad.float(), xr.grad, rtol=2 ** -7, atol=2e-2)
    scale = wr.grad.abs().max().item()
    assert (w.grad - wr.grad).abs().max().item() < 1e-2 * scale
    assert torch.allclose(b.grad, br.grad, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k", [(384, 256), (384, 128), (384, 64), (192, 64), (192, 256), (128, 256), (128, 128),
                                 (128, 64), (64, 256), (64, 128), (64, 64)])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 70_001])
def test_linear_wgrad_matches_fp64(cuda, dtype, m, k, n):
    """segger_linear_wgrad: dW = dY^T X and db = sum dY in one pass (MFMA, fp32 accumulate) against float64 on the
    same rounded inputs.  Error of an fp32 sum of n products of magnitude ~1: <= 1e-5 of sum |dY| |X| (+ the same for
    the column sums); exact integers check the operand maps element by element."""
    from segger_amd import ops
    assert ops.linear_wgrad_supported(m, k, dtype)
    g = torch.Generator(device=cuda).manual_seed(n + m + k)
    gy = (torch.randn(n, m, device=cuda, generator=g) + 0.1).to(dtype)
    x = torch.randn(n, k, device=cuda, generator=g).to(dtype)
    gw, gb = ops.linear_wgrad_launch(gy, x)
    assert gw.shape == (m, k) and gb.shape == (m,) and gw.dtype == gb.dtype == torch.float32
    ref_w = gy.double().t() @ x.double()
    ref_b = gy.double().sum(0)
    bound_w = 1e-5 * (gy.double().abs().t() @ x.double().abs()) + 1e-6
    assert bool(((gw.double() - ref_w).abs() <= bound_w).all())
    assert bool(((gb.double() - ref_b).abs() <= 1e-5 * gy.double().abs().sum(0) + 1e-6).all())
    gw2, gb2 = ops.linear_wgrad_launch(gy, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)                # deterministic
    gw3, none = ops.linear_wgrad_launch(gy, x, want_bias=False)
    assert none is None and torch.equal(gw3, gw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m", [384, 192, 128, 64])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 31, 33, 1000, 8193, 70_001])
def test_linear_wgrad_dx_matches_fp64(cuda, dtype, m, n):
    """segger_linear_wgrad_dx: dX = dY W, dW = dY^T X and db = sum dY in ONE pass over dY (K = 128) against float64 on
    the same rounded inputs.  dW / db: as test_linear_wgrad_matches_fp64 (and bit-identical to the separate kernel: the
    same accumulation order).  dX: fp32 accumulation of M <= 384 products, one rounding to the storage dtype."""
    from segger_amd import ops
    k = 128
    assert ops.linear_wgrad_dx_supported(m, k, dtype) and not ops.linear_wgrad_dx_supported(m, 256, dtype)
    g = torch.Generator(device=cuda).manual_seed(n + m)
    gy = (torch.randn(n, m, device=cuda, generator=g) + 0.1).to(dtype)
    x = torch.randn(n, k, device=cuda, generator=g).to(dtype)
    w = (torch.randn(m, k, device=cuda, generator=g) / m ** 0.5).to(dtype)
    gx, gw, gb = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous())
    assert gx.shape == (n, k) and gx.dtype == dtype and gw.shape == (m, k) and gb.shape == (m,)
    ref_w = gy.double().t() @ x.double()
    bound_w = 1e-5 * (gy.double().abs().t() @ x.double().abs()) + 1e-6
    assert bool(((gw.double() - ref_w).abs() <= bound_w).all())
    assert bool(((gb.double() - gy.double().sum(0)).abs() <= 1e-5 * gy.double().abs().sum(0) + 1e-6).all())
    gw1, gb1 = ops.linear_wgrad_launch(gy, x)
    assert torch.equal(gw, gw1) and torch.equal(gb, gb1)
    ref_x = gy.double() @ w.double()
    rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert bool(((gx.double() - ref_x).abs() <= rel * ref_x.abs() + 1e-3).all())
    gx2, gw2, gb2 = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous())
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)   # deterministic
    gx3, gw3, none = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous(), want_bias=False)
    assert none is None and torch.equal(gx3, gx) and torch.equal(gw3, gw)


def test_linear_wgrad_dx_operand_maps_exact_and_strided(cuda):
    """Small exact integers: dX and dW must equal the integer matmuls bit for bit, for asymmetric patterns, with dY / X
    given as column windows of wider matrices, over a row count that leaves a partial last stage and several slabs."""
    from segger_amd import ops
    for n in (37, 16 * 32 * 3 + 5):
        m, k = 384, 128
        r = torch.arange(n, device=cuda)
        big_y = ((r[:, None] * 3 + torch.arange(3 * m, device=cuda)[None] * 5) % 7 - 3).to(torch.bfloat16)
        big_x = ((r[:, None] * 2 + torch.arange(2 * k, device=cuda)[None] * 11) % 5 - 2).to(torch.bfloat16)
        w = ((torch.arange(m, device=cuda)[:, None] * 7 + torch.arange(k, device=cuda)[None] * 3) % 5 - 2).to(torch.bfloat16)
        gy, x = big_y[:, m:2 * m], big_x[:, k:]
        gx, gw, gb = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous())
        assert torch.equal(gw, gy.float().t() @ x.float())
        assert torch.equal(gb, gy.float().sum(0))
        want = gy.float() @ w.float()                       # |values| <= 3 * 2 * 384: exact in fp32, and in bf16 up to 256
        assert torch.equal(gx.float(), want.to(torch.bfloat16).float())
    gx0, gw0, gb0 = ops.linear_wgrad_dx_launch(gy[:0], x[:0], w.t().contiguous())
    assert gx0.shape == (0, k) and not gw0.any() and not gb0.any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m", [384, 128])
@pytest.mark.parametrize("n", [1, 17, 1000, 70_001])
def test_linear_wgrad_dx_gelu_gate(cuda, dtype, m, n):
    """segger_linear_wgrad_dx with a gate: dX = (dY W) * gelu'(gate) from the fp32 accumulator (one rounding), dW / db
    bit-identical to the ungated launch; gate given as a column window of a wider matrix."""
    from segger_amd import ops
    k = 128
    assert ops.linear_wgrad_dx_gate_supported(m, k, dtype) and not ops.linear_wgrad_dx_gate_supported(192, k, dtype)
    g = torch.Generator(device=cuda).manual_seed(3 * n + m)
    gy = torch.randn(n, m, device=cuda, generator=g).to(dtype)
    x = torch.randn(n, k, device=cuda, generator=g).to(dtype)
    w = (torch.randn(m, k, device=cuda, generator=g) / m ** 0.5).to(dtype)
    wide = (2.0 * torch.randn(n, 2 * k + 8, device=cuda, generator=g)).to(dtype)
    gate = wide[:, 8:8 + k]
    gx, gw, gb = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous(), gate=gate)
    gx0, gw0, gb0 = ops.linear_wgrad_dx_launch(gy, x, w.t().contiguous())
    assert torch.equal(gw, gw0) and torch.equal(gb, gb0)
    z = gate.double()
    dgelu = 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5
    ref = (gy.double() @ w.double()) * dgelu
    rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert bool(((gx.double() - ref).abs() <= rel * ref.abs() + 1e-3).all())
    with pytest.raises(RuntimeError):                          # m_out = 192: no room for the gate rows in its stage ring
        ops.linear_wgrad_dx_launch(gy.new_zeros(n, 192), x, w.new_zeros(k, 192), gate=gate)


@pytest.mark.parametrize("m", [384, 64])
def test_linear_autograd_fused_equals_separate_kernels(cuda, m):
    """ops.linear's backward through the one-pass kernel == through the separate dX GEMM + weight-gradient kernel:
    parameter gradients bit for bit (same kernel arithmetic), dX within one rounding of the storage dtype."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(m)
    n, k = 50_003, 128
    x = torch.randn(n, k, device=cuda, generator=g).to(torch.bfloat16)
    w = torch.nn.Parameter(torch.randn(m, k, device=cuda, generator=g) / k ** 0.5)
    b = torch.nn.Parameter(torch.randn(m, device=cuda, generator=g))
    gy = torch.randn(n, m, device=cuda, generator=g).to(torch.bfloat16)
    out = {}
    for fused in (True, False):
        ops.FUSED_WGRAD_DX = fused
        try:
            xi = x.clone().requires_grad_(True)
            w.grad = b.grad = None
            ops.linear(xi, w, b).backward(gy)
            out[fused] = (xi.grad.clone(), w.grad.clone(), b.grad.clone())
        finally:
            ops.FUSED_WGRAD_DX = True
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
    a, r = out[True][0].float(), out[False][0].float()
    assert bool(((a - r).abs() <= 2.0 ** -7 * r.abs() + 1e-3).all())


def test_linear_wgrad_operand_maps_exact_and_strided(cuda):
    """Small exact integers (every product and sum representable): dW must equal the integer matmul bit for bit, for
    an asymmetric pattern, with dY / X given as column windows of wider matrices (row stride != width)."""
    from segger_amd import ops
    n, m, k = 37, 384, 256
    r = torch.arange(n, device=cuda)
    big_y = ((r[:, None] * 3 + torch.arange(3 * m, device=cuda)[None] * 5) % 7 - 3).to(torch.bfloat16)
    big_x = ((r[:, None] * 2 + torch.arange(2 * k, device=cuda)[None] * 11) % 5 - 2).to(torch.bfloat16)
    gy, x = big_y[:, m:2 * m], big_x[:, k:]
    gw, gb = ops.linear_wgrad_launch(gy, x)
    assert torch.equal(gw, gy.float().t() @ x.float())
    assert torch.equal(gb, gy.float().sum(0))
    gw0, gb0 = ops.linear_wgrad_launch(gy[:0], x[:0])
    assert not gw0.any() and not gb0.any()


@pytest.mark.parametrize("fused", [True, False])
def test_cached_weight_copies_follow_the_optimizer(cuda, fused):
    """The compute-dtype weight copies are cached between forwards; a FUSED optimizer updates parameters in place
    without bumping their version counters, so the cache is keyed on the optimizer-step generation as well: after
    every step the projection must use the new weights (a stale copy trains nothing -- caught by the FOV AUROC run)."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(0)
    w1 = torch.nn.Parameter(torch.randn(64, 128, device=cuda, generator=g) / 11)
    w2 = torch.nn.Parameter(torch.randn(64, 128, device=cuda, generator=g) / 11)
    b1 = torch.nn.Parameter(torch.zeros(64, device=cuda))
    x = torch.randn(500, 128, device=cuda, generator=g).to(torch.bfloat16)
    opt = torch.optim.Adam([w1, w2, b1], lr=0.05, fused=fused, foreach=not fused)
    for _ in range(3):
        y = ops.linear(x, (w1, w2), (b1, None))
        ref = x.float() @ torch.cat([w1, w2]).detach().to(torch.bfloat16).float().t() + torch.cat([b1.detach(), b1.new_zeros(64)])
        assert torch.allclose(y.float(), ref, rtol=2 ** -7, atol=2e-2)
        opt.zero_grad()
        y.float().pow(2).mean().backward()
        before = w1.detach().clone()
        opt.step()
        assert not torch.equal(before, w1.detach())
    # the refresh itself (one launch, segger_pack_refresh): stacked 16-bit copy, its transpose, the stacked fp32 bias
    xg = x.clone().requires_grad_(True)
    ops.linear(xg, (w1, w2), (b1, None)).float().sum().backward()          # (the data gradient makes the transposed copy)
    opt.step()
    ops.linear(x, (w1, w2), (b1, None))
    pk = ops._pack_for((w1, w2), (b1, None))
    want = torch.cat([w1, w2]).detach().to(torch.bfloat16)
    assert torch.equal(pk.w, want) and torch.equal(pk.b, torch.cat([b1.detach(), b1.new_zeros(64)]))
    assert pk._wt is not None and pk._wt_fresh and torch.equal(pk._wt, want.t().contiguous())
    with torch.no_grad():                                # parameters written behind autograd's back need the explicit call
        w1.data.mul_(2.0)
    ops.invalidate_weight_cache()
    y = ops.linear(x, (w1, w2), (b1, None))
    assert torch.allclose(y[:, :64].float(), x.float() @ w1.detach().to(torch.bfloat16).float().t() + b1.detach(), rtol=2 ** -7, atol=2e-2)


def test_default_model_projections_are_all_on_the_mfma_kernels_at_every_width(cuda):
    """CLI-default dims (in 128, hidden / out 64, 2 heads: reference lightning_model.py:30-35, segment.py:201-235): every
    projection of a training step -- forward, data gradient, weight gradient -- is covered by the hand-written kernels
    at bf16, f16 AND fp32 storage (the fp32 model-level parity tests of test_gpu_model.py therefore run on
    v_mfma_f32_32x32x2_f32, not on the vendor GEMM)."""
    from segger_amd import ops
    fwd = [(256, 384), (128, 384), (128, 128), (256, 128), (128, 64), (256, 64), (64, 64)]     # (K, M)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for k, m in fwd:
            assert ops.linear_supported(k, m, dt), (k, m, dt)
            assert ops.linear_supported(m, k, dt), ("dX", m, k, dt)
            assert ops.linear_wgrad_supported(m, k, dt), ("dW", m, k, dt)


def test_unsupported_shapes_use_vendor_gemm(cuda):
    from segger_amd import ops
    assert not ops.linear_supported(100, 64, torch.bfloat16)
    assert not ops.linear_supported(128, 96, torch.bfloat16)
    assert ops.linear_supported(128, 64, torch.float32)          # fp32 storage: the exact-fp32 MFMA kernels (round 3)
    x = torch.randn(10, 100, device=cuda).to(torch.bfloat16)
    w = torch.randn(96, 100, device=cuda)
    assert ops.linear(x, w, None).shape == (10, 96)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,cols", [(1, 64), (255, 8), (4097, 384), (100_003, 128), (70_001, 2048), (600_000, 64)])
def test_colsum_matches_fp64_sum(cuda, dtype, n, cols):
    """Bias-gradient reduction: fp32 accumulation of <= 6e5 values of magnitude ~1 -> rtol 1e-5 of sum |x|."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(n + cols)
    x = (torch.randn(n, cols, device=cuda, generator=g) + 0.25).to(dtype)
    got = ops.colsum(x)
    ref = x.double().sum(0)
    bound = 1e-5 * x.double().abs().sum(0) + 1e-6
    assert got.dtype == torch.float32 and got.shape == (cols,)
    assert bool(((got.double() - ref).abs() <= bound).all())
    # a column window of a wider matrix (row stride != cols), as the fused projection gradient is laid out
    if cols >= 16:
        view = x[:, cols // 2:]
        assert bool(((ops.colsum(view).double() - ref[cols // 2:]).abs() <= bound[cols // 2:]).all())
    assert torch.equal(ops.colsum(x), got)                  # deterministic


def test_colsum_empty_and_errors(cuda):
    from segger_amd import _lib, ops
    assert torch.equal(ops.colsum(torch.empty(0, 64, device=cuda)), torch.zeros(64, device=cuda))
    lib = _lib.load()
    x = torch.zeros(10, 64, device=cuda)
    out = torch.zeros(64, device=cuda)
    rc = lib.segger_colsum(x.data_ptr(), 64, 10, 64, 0, out.data_ptr(), None, 0, _lib.stream_ptr(cuda))
    assert rc != 0 and b"workspace" in lib.segger_last_error()
    rc = lib.segger_colsum(x.data_ptr(), 64, 10, 60, 0, out.data_ptr(), None, 0, _lib.stream_ptr(cuda))
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("k,ma,mb,na,nb", [(128, 384, 128, 3000, 45), (256, 384, 128, 1001, 130), (128, 64, 64, 257, 7),
                                            (128, 384, 128, 400_000, 500)])
def test_linear_pair_equals_two_launches(cuda, dtype, k, ma, mb, na, nb):
    """segger_linear_fwd_pair (a hetero layer's transcript and boundary projections, lin_last of both node types, as one
    launch) against two ops.linear calls: outputs bit-identical, all gradients equal (the same backward kernels run).
    The last case sends the large side to its persistent resident-W launch."""
    from segger_amd import ops
    g = torch.Generator().manual_seed(k + ma + na)
    xa0 = torch.randn(na, k, generator=g).to(dtype)
    xb0 = torch.randn(nb, k, generator=g).to(dtype)
    ws = [torch.randn(m, k, generator=g) * 0.05 for m in ([ma // 3] * 3 if ma == 384 else [ma]) + [mb]]
    bs = [torch.randn(w.shape[0], generator=g) for w in ws]
    out = {}
    for pair in (True, False):
        xa, xb = xa0.to(cuda).requires_grad_(True), xb0.to(cuda).requires_grad_(True)
        w = [t.to(cuda).requires_grad_(True) for t in ws]
        b = [t.to(cuda).requires_grad_(True) for t in bs]
        if pair:
            ya, yb = ops.linear_pair(xa, w[:-1], b[:-1], xb, w[-1], b[-1])
        else:
            ya, yb = ops.linear(xa, w[:-1], b[:-1]), ops.linear(xb, w[-1], b[-1])
        assert ya.shape == (na, ma) and yb.shape == (nb, mb)
        (ya.float().square().sum() + yb.float().sum()).backward()
        out[pair] = [ya.detach(), yb.detach(), xa.grad, xb.grad] + [t.grad for t in w] + [t.grad for t in b]
    for p_, s_ in zip(out[True], out[False]):
        assert torch.equal(p_, s_)


@pytest.mark.parametrize("k,m,n", [(128, 384, 4133), (128, 64, 1), (128, 128, 129), (384, 128, 3001), (128, 384, 40037), (384, 128, 40037),
                                   (384, 128, 31), (128, 384, 8192)])
def test_linear_fp32_split_within_the_exact_kernels_error(cuda, k, m, n):
    """segger_linear_fwd_f32_split (fp32 operands as three bf16 parts, six partial products on the bf16 MFMA, fp32
    accumulation) against fp64: its error, relative to sum |x||w| (what fp32 rounding scales with), stays within twice the
    exact-fp32 MFMA kernel's own and within 4 * 2^-24 * sqrt(K) absolutely; partial last tile, bias, wide dynamic range."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(k + m + n)
    x = torch.randn(n, k, device=cuda, generator=g) * torch.rand(n, 1, device=cuda, generator=g).mul(6).exp()
    w = torch.randn(m, k, device=cuda, generator=g) / k ** 0.5
    b = torch.randn(m, device=cuda, generator=g)
    assert ops.linear_f32_split_supported(k, m)
    w3 = ops.f32_split_planes(w)
    assert torch.equal(w3.float().sum(0), w)                  # the three parts add up to the fp32 number exactly
    assert torch.equal(w3.cpu(), ops.f32_split_planes(w.cpu()))                                   # kernel == the torch formulas
    assert torch.equal(ops.f32_split_planes(w, transposed=True), ops.f32_split_planes(w.t().contiguous()))
    y_split = ops.linear_f32_split_launch(x, w3, b)
    y_exact = ops.linear_fwd_launch(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    bound = x.double().abs() @ w.double().abs().t() + b.double().abs()
    e_split = ((y_split.double() - ref).abs() / bound).max().item()
    e_exact = ((y_exact.double() - ref).abs() / bound).max().item()
    assert e_split <= max(2 * e_exact, 2.0 ** -23), (e_split, e_exact)
    assert e_split <= 4 * 2.0 ** -24 * k ** 0.5


@pytest.mark.parametrize("k,m,gate", [(128, 384, None), (384, 128, None), (384, 128, "gelu")])
def test_linear_fp32_w_resident_kernels_at_c2_size(cuda, k, m, gate):
    """The W-resident split kernels at BASELINE's C2 row count (1M rows: every persistent workgroup walks ~122 tiles, the
    deferred stores / refilled row registers / gate prefetch in steady state) against fp64 on sampled rows plus the first
    and the last tile: error relative to sum |x||w| within the exact-fp32 kernel's class (it measures 3.5e-7 here; bar 2^-21)."""
    from segger_amd import ops
    n = 1_000_003                                            # a partial last tile
    g = torch.Generator(device=cuda).manual_seed(k + m)
    x = torch.randn(n, k, device=cuda, generator=g) * torch.rand(n, 1, device=cuda, generator=g).mul(4).exp()
    w = torch.randn(m, k, device=cuda, generator=g) / k ** 0.5
    idx = torch.cat([torch.arange(0, 64, device=cuda), torch.arange(n - 64, n, device=cuda),
                     torch.randint(0, n, (4096,), device=cuda, generator=g)])
    if gate is None:
        b = torch.randn(m, device=cuda, generator=g)
        y = ops.linear_f32_split_launch(x, ops.f32_split_planes(w), b)
        ref = x[idx].double() @ w.double().t() + b.double()
        bound = x[idx].double().abs() @ w.double().abs().t() + b.double().abs()
    else:
        gt = torch.randn(n, m, device=cuda, generator=g)
        y = ops.linear_f32_gate_launch(x, w, gt, gate)
        dg = torch.ops.aten.gelu_backward(torch.ones(len(idx), m, device=cuda, dtype=torch.float64), gt[idx].double())
        ref = (x[idx].double() @ w.double().t()) * dg
        bound = x[idx].double().abs() @ w.double().abs().t()       # (not times |gelu'|: it has a zero, and is evaluated in fp32)
    err = ((y[idx].double() - ref).abs() / bound).max().item()
    assert err <= 2.0 ** -21 * (1 if gate is None else 2), err      # (gelu' itself is evaluated in fp32)
    assert bool(torch.isfinite(y).all())


def test_linear_fp32_split_autograd_switch(cuda, monkeypatch):
    """ops.F32_SPLIT routes the fp32 forward projection and its data gradient through the split kernel; outputs and all
    gradients agree with the exact route to fp32 rounding."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(9)
    n, k = 5000, 128
    x0 = torch.randn(n, k, device=cuda, generator=g)
    ws = [torch.randn(128, k, device=cuda, generator=g) / k ** 0.5 for _ in range(3)]
    bs = [torch.randn(128, device=cuda, generator=g) for _ in range(3)]
    gy = torch.randn(n, 384, device=cuda, generator=g)
    out = {}
    for split in (False, True):
        monkeypatch.setattr(ops, "F32_SPLIT", split)
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        y = ops.linear(x, w, b)
        y.backward(gy)
        out[split] = [y.detach(), x.grad] + [t.grad for t in w] + [t.grad for t in b]
    for a, r in zip(out[True], out[False]):
        assert (a - r).abs().max().item() <= 2e-5 * r.abs().max().item()
    assert not torch.equal(out[True][0], out[False][0])       # (it did take the other kernel)


@pytest.mark.parametrize("m,k,n", [(384, 128, 40037), (384, 128, 17), (128, 128, 5000), (64, 256, 8191), (64, 64, 3), (64, 128, 1024),
                                   (128, 256, 2049)])
def test_wgrad_fp32_split_within_the_exact_kernels_error(cuda, m, k, n, monkeypatch):
    """segger_linear_wgrad_f32_split (dW = dY^T X, db = sum dY; both fp32 operands as three bf16 parts, six partial products
    on the bf16 MFMA, fp32 accumulation) against fp64: the error relative to sum |dY||X| stays within twice the exact-fp32
    kernel's own; partial last stage, a single row, row strides (views into wider matrices), wide dynamic range."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(m + k + n)
    gy_w = torch.randn(n, m + 64, device=cuda, generator=g) * torch.rand(n, 1, device=cuda, generator=g).mul(5).exp()
    x_w = torch.randn(n, k + 32, device=cuda, generator=g)
    gy, x = gy_w[:, :m], x_w[:, :k]
    out = {}
    for split in (True, False):
        monkeypatch.setattr(ops, "F32_SPLIT", split)
        out[split] = ops.linear_wgrad_launch(gy, x)
    ref_w = gy.double().t() @ x.double()
    ref_b = gy.double().sum(0)
    bound = gy.double().abs().t() @ x.double().abs()
    e = {s_: ((out[s_][0].double() - ref_w).abs() / bound).max().item() for s_ in out}
    assert e[True] <= max(2 * e[False], 2.0 ** -22), e
    eb = {s_: ((out[s_][1].double() - ref_b).abs() / gy.double().abs().sum(0)).max().item() for s_ in out}
    assert eb[True] <= max(2 * eb[False], 2.0 ** -21), eb
    from segger_amd import _lib
    took_split = bool(_lib.load().segger_linear_wgrad_f32_split_supported(m, k))     # ((64, 64) / (64, 128): the exact kernel is faster)
    if n > 16:
        assert torch.equal(out[True][0], out[False][0]) != took_split      # (it did take the other kernel)


@pytest.mark.parametrize("m,k", [(384, 128), (64, 256)])
def test_wgrad_fp32_split_at_c2_size(cuda, m, k):
    """The software-pipelined split weight gradient at 1M rows (every workgroup runs ~244 stages: the ring refill and the
    two LDS buffers in steady state, a partial last stage) against fp64."""
    from segger_amd import ops
    n = 1_000_003
    g = torch.Generator(device=cuda).manual_seed(m * k)
    gy = torch.randn(n, m, device=cuda, generator=g)
    x = torch.randn(n, k, device=cuda, generator=g)
    gw, gb = ops.linear_wgrad_launch(gy, x)
    ref_w = gy.double().t() @ x.double()
    bound = gy.double().abs().t() @ x.double().abs()
    assert ((gw.double() - ref_w).abs() / bound).max().item() <= 2.0 ** -24
    assert ((gb.double() - gy.double().sum(0)).abs() / gy.double().abs().sum(0)).max().item() <= 2.0 ** -22


@pytest.mark.parametrize("kind", ["gelu", "silu"])
@pytest.mark.parametrize("k,m,n,split", [(384, 128, 3001, True), (384, 128, 3001, False), (64, 64, 4097, False), (128, 128, 130, True),
                                         (256, 64, 77, False), (384, 128, 40037, True), (384, 128, 1, True)])
def test_linear_f32_gate_epilogue(cuda, k, m, n, split, kind, monkeypatch):
    """segger_linear_fwd_f32_gate: (x @ W^T) * act'(gate) in one kernel (exact-fp32 MFMA or the bf16x3 split) against torch's
    matmul + gelu_backward / silu_backward in float64; gate and x as views with a row stride."""
    from segger_amd import ops
    monkeypatch.setattr(ops, "F32_SPLIT", split)
    if not split and k == 384:
        pytest.skip("the exact kernel's gate form covers k_in 64 / 128 / 256 (the split serves 384)")
    g = torch.Generator(device=cuda).manual_seed(k + m + n)
    xw = torch.randn(n, k + 16, device=cuda, generator=g)
    gw = torch.randn(n, m + 8, device=cuda, generator=g) * 2
    x, gate = xw[:, :k], gw[:, :m]
    w = torch.randn(m, k, device=cuda, generator=g) / k ** 0.5
    y = ops.linear_f32_gate_launch(x, w, gate, kind)
    gd = gate.double().requires_grad_(True)
    act = torch.nn.functional.gelu(gd) if kind == "gelu" else torch.nn.functional.silu(gd)
    (dact,) = torch.autograd.grad(act.sum(), gd)
    ref = (x.double() @ w.double().t()) * dact
    # (relative to sum |x||w|, not to |act'|: near the zero of gelu' the epilogue's erf-free derivative carries its ~1e-6
    #  absolute error into a product whose own size vanishes)
    bound = x.double().abs() @ w.double().abs().t()
    assert ((y.double() - ref).abs() / (bound + 1e-6)).max().item() < 2e-5
