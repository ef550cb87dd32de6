"""Parity of CSR build, prediction head and triplet head (C ABI) with the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_csr_from_coo_matches_stable_sort(cuda):
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(1)
    for n_rows, n_cols, E in [(1, 1, 1), (10, 7, 0), (37, 91, 1000), (100_000, 1000, 250_000), (5, 5, 4096)]:
        row = torch.randint(0, n_rows, (E,), generator=g)
        col = torch.randint(0, n_cols, (E,), generator=g)
        csr = csr_from_coo(row.to(cuda), col.to(cuda), n_rows, n_cols)
        order = torch.argsort(row, stable=True)
        counts = torch.bincount(row, minlength=n_rows)
        indptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
        assert torch.equal(csr.indptr.cpu(), indptr)
        assert torch.equal(csr.eid.cpu().long(), order)
        assert torch.equal(csr.col.cpu().long(), col[order])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 32, 48, 20])
def test_edge_cos_argmax(oracle, cuda, dtype, C):
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(C)
    n_tx, n_bd, E = 500, 40, 1300
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, C, generator=g), dim=-1).to(dtype)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, C, generator=g), dim=-1).to(dtype)
    src = torch.randint(5, n_tx, (E,), generator=g)          # tx 0..4 have no candidate
    dst = torch.randint(0, n_bd, (E,), generator=g)
    # exact ties: duplicate some edges (same src, same dst) later in the list
    src = torch.cat([src, src[:50]]); dst = torch.cat([dst, dst[:50]])
    ei = torch.stack([src, dst])
    bd_index = torch.randperm(n_bd, generator=g).to(torch.int32) + 1000
    for min_sim in (None, 0.1):
        seg_ref, max_ref = oracle.predict_assign(z_tx.double(), z_bd.double(), ei, bd_index, min_sim)
        sim_ref = oracle.edge_scores(z_tx.double(), z_bd.double(), ei)
        by_src = csr_from_coo(src.to(cuda), dst.to(cuda), n_tx, n_bd)
        max_sim, max_eid, seg, sim = ops.edge_cos_argmax(by_src, z_tx.to(cuda), z_bd.to(cuda), dst_index=bd_index.to(cuda),
                                                         min_similarity=min_sim, return_sim=True)
        tol = 2e-6 if dtype == torch.float32 else 1e-5      # inputs identical; fp32 dot of 64 terms
        assert torch.allclose(sim.cpu().double(), sim_ref, atol=tol)
        assert torch.allclose(max_sim.cpu().double(), max_ref, atol=tol)
        # the arg-max may legitimately differ from float64 only where two candidates are within tol
        _, arg_ref = oracle.scatter_max(sim.cpu(), src, n_tx)        # ties -> lowest edge id, on the HIP scores
        assert torch.equal(max_eid.cpu(), arg_ref)
        assert (max_eid[:5].cpu() == ei.shape[1]).all() and (seg[:5].cpu() == -1).all() and (max_sim[:5].cpu() == 0).all()
        agree = (seg.cpu() == seg_ref).float().mean()
        assert agree > 0.995, f"assignment agreement {agree}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_triplet_edge_loss(oracle, cuda, dtype):
    from segger_amd import ops
    g = torch.Generator().manual_seed(5)
    n_tx, n_bd, E, C = 700, 30, 2000, 64
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, C, generator=g), dim=-1).to(dtype)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, C, generator=g), dim=-1).to(dtype)
    src = torch.randint(0, n_tx, (E,), generator=g)
    dst = torch.randint(0, n_bd, (E,), generator=g)
    neg = (dst + torch.randint(1, n_bd, (E,), generator=g)) % n_bd
    a, b = z_tx.double().requires_grad_(True), z_bd.double().requires_grad_(True)
    ref = oracle.segmentation_loss(a, b, torch.stack([src, dst]), neg, "triplet", 0.4)
    (ref * 0.37).backward()
    da, db = z_tx.to(cuda).requires_grad_(True), z_bd.to(cuda).requires_grad_(True)
    loss = ops.triplet_edge_loss(da, db, src.to(cuda), dst.to(cuda), neg.to(cuda), 0.4)
    (loss * 0.37).backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    # fp32: atomics reorder fp32 sums; bf16: the returned gradient is rounded to bf16 (2^-9 relative)
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float32 else (1e-2, 2e-6)
    assert torch.allclose(da.grad.cpu().double(), a.grad, rtol=rtol, atol=atol)
    assert torch.allclose(db.grad.cpu().double(), b.grad, rtol=rtol, atol=atol)


def test_masked_tx_triplet_loss_equals_gathered_form(cuda):
    """loss_tx through the fused kernel (anchors / positives / negatives index one matrix) equals
    TripletLoss.forward on the gathered embeddings, values and gradients."""
    from segger_amd.triplet_loss import TripletLoss
    g = torch.Generator().manual_seed(9)
    n, c, k = 5000, 64, 6
    a = torch.randn(k, 4, generator=g); a = a / a.norm(dim=1, keepdim=True)
    sim = a @ a.t()
    z = torch.nn.functional.normalize(torch.randn(n, c, generator=g), dim=-1)
    labels = torch.randint(0, k, (n,), generator=g)
    mask = torch.rand(n, generator=g) < 0.8
    loss_fn = TripletLoss(sim, margin=0.3)
    zd = z.to(cuda).requires_grad_(True)
    torch.manual_seed(77)
    l1 = loss_fn.forward_masked(zd, labels.to(cuda), mask.to(cuda))
    (l1 * 0.5).backward()
    zr = z.to(cuda).requires_grad_(True)
    torch.manual_seed(77)
    l2 = loss_fn.forward(zr[mask.to(cuda)], labels.to(cuda)[mask.to(cuda)])
    (l2 * 0.5).backward()
    assert abs(l1.item() - l2.item()) < 1e-6
    assert torch.allclose(zd.grad, zr.grad, atol=1e-7, rtol=1e-4)
