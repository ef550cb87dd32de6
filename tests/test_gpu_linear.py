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


def test_unsupported_shapes_use_vendor_gemm(cuda):
    from segger_amd import ops
    assert not ops.linear_supported(100, 64, torch.bfloat16)
    assert not ops.linear_supported(128, 96, torch.bfloat16)
    assert not ops.linear_supported(128, 64, torch.float32)
    x = torch.randn(10, 100, device=cuda).to(torch.bfloat16)
    w = torch.randn(96, 100, device=cuda)
    assert ops.linear(x, w, None).shape == (10, 96)
