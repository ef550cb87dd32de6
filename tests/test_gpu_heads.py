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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [64, 32, 21])           # 64: vectorised kernel; 32: pair atomics; 21: scalar fp32 atomics
@pytest.mark.parametrize("boundary_side", ["atomics", "packed_anchor", "segment_sum"])
def test_triplet_edge_loss(oracle, cuda, dtype, C, boundary_side, monkeypatch):
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    # "packed_anchor": the anchor side accumulates with packed 16-bit atomics (large edge lists); "segment_sum": the
    # positives of the boundary side are a segmented sum over the triplets grouped by positive row, no atomics
    monkeypatch.setattr(ops, "_CONTRIB_MIN_EDGES", 1 << 60 if boundary_side == "atomics" else 0)
    g = torch.Generator().manual_seed(5)
    n_tx, n_bd, E = 700, 30, 2000
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, C, generator=g), dim=-1).to(dtype)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, C, generator=g), dim=-1).to(dtype)
    src = torch.randint(0, n_tx, (E,), generator=g)
    dst = torch.randint(0, n_bd, (E,), generator=g)
    dst[dst == 7] = 8                                    # a boundary that is nobody's positive
    neg = (dst + torch.randint(1, n_bd, (E,), generator=g)) % n_bd
    a, b = z_tx.double().requires_grad_(True), z_bd.double().requires_grad_(True)
    ref = oracle.segmentation_loss(a, b, torch.stack([src, dst]), neg, "triplet", 0.4)
    (ref * 0.37).backward()
    da, db = z_tx.to(cuda).requires_grad_(True), z_bd.to(cuda).requires_grad_(True)
    groups = None
    if boundary_side == "segment_sum":          # triplets grouped by positive row (boundary 7 gets none: dst != 7 below)
        groups = csr_from_coo(dst.to(cuda), src.to(cuda), n_bd, n_tx, validate=False)
    loss = ops.triplet_edge_loss(da, db, src.to(cuda), dst.to(cuda), neg.to(cuda), 0.4, pos_groups=groups)
    (loss * 0.37).backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    # fp32: atomics reorder fp32 sums; 16-bit: the anchor-side gradient is accumulated in the embedding dtype
    # (packed atomics, ~3 terms per row here), the boundary side in fp32 and rounded once
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float32 else (1e-2, 2e-6)
    assert torch.allclose(da.grad.cpu().double(), a.grad, rtol=rtol, atol=atol)
    assert torch.allclose(db.grad.cpu().double(), b.grad, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [64, 32, 128])
@pytest.mark.parametrize("packed", [False, True])
def test_triplet_backward_in_one_walk_over_the_groups(oracle, cuda, dtype, C, packed, monkeypatch):
    """``anchors_unique`` (tx-belongs-bd: a transcript lies in at most one boundary): segger_triplet_bwd walks the
    triplets grouped by positive row once -- anchor rows stored (fp32, or packed pairs in the embeddings' dtype),
    positives summed in registers, negatives by fp32 atomics -- against the float64 oracle; a transcript without an
    edge keeps a zero gradient; a boundary that is nobody's positive still collects its negatives."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    monkeypatch.setattr(ops, "_CONTRIB_MIN_EDGES", 0 if packed else 1 << 60)
    g = torch.Generator().manual_seed(11 + C)
    n_tx, n_bd = 3000, 40
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, C, generator=g), dim=-1).to(dtype)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, C, generator=g), dim=-1).to(dtype)
    src = torch.randperm(n_tx, generator=g)[:2500]                  # unique anchors; 500 transcripts without an edge
    dst = torch.randint(0, n_bd, (2500,), generator=g)
    dst[dst == 7] = 8
    neg = (dst + torch.randint(1, n_bd, (2500,), generator=g)) % n_bd
    a, b = z_tx.double().requires_grad_(True), z_bd.double().requires_grad_(True)
    ref = oracle.segmentation_loss(a, b, torch.stack([src, dst]), neg, "triplet", 0.4)
    (ref * 0.37).backward()
    da, db = z_tx.to(cuda).requires_grad_(True), z_bd.to(cuda).requires_grad_(True)
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), n_bd, n_tx, validate=False)
    asked = []
    loss = ops.triplet_edge_loss(da, db, src.to(cuda), dst.to(cuda), neg.to(cuda), 0.4, pos_groups=groups,
                                 anchors_unique=lambda: asked.append(1) or True)
    (loss * 0.37).backward()
    assert asked == [1]                                             # asked in the backward, once
    assert abs(loss.item() - ref.item()) < 1e-5
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float32 else (1e-2, 2e-6)
    assert torch.allclose(da.grad.cpu().double(), a.grad, rtol=rtol, atol=atol)
    assert torch.allclose(db.grad.cpu().double(), b.grad, rtol=rtol, atol=atol)
    free = torch.ones(n_tx, dtype=torch.bool); free[src] = False
    assert not da.grad.cpu()[free].any()
    # the two-kernel route (anchors not known to be unique) gives the same gradients
    d2, b2 = z_tx.to(cuda).requires_grad_(True), z_bd.to(cuda).requires_grad_(True)
    (ops.triplet_edge_loss(d2, b2, src.to(cuda), dst.to(cuda), neg.to(cuda), 0.4, pos_groups=groups) * 0.37).backward()
    assert torch.allclose(d2.grad.float(), da.grad.float(), rtol=rtol, atol=atol)
    assert torch.allclose(b2.grad.float(), db.grad.float(), rtol=rtol, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [64, 32, 22])
@pytest.mark.parametrize("route", ["atomics", "grouped"])
def test_bce_edge_loss(oracle, cuda, dtype, C, route, monkeypatch):
    """The BCE variant of the segmentation loss (lightning_model.py:190-207) as one kernel each way: BCEWithLogits over
    the positive / sampled-negative dot products, against the float64 oracle; "grouped": unique anchors, one walk over
    the by-destination groups (anchor rows stored); C = 22: not a multiple of 32 -> the per-edge kernel with atomics."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    monkeypatch.setattr(ops, "_CONTRIB_MIN_EDGES", 0 if dtype != torch.float32 and route == "grouped" else 1 << 60)
    g = torch.Generator().manual_seed(3 + C)
    n_tx, n_bd, E = 2600, 35, 2000
    z_tx = torch.randn(n_tx, C, generator=g).to(dtype)          # (not normalised: logits of a few units either way)
    z_bd = (torch.randn(n_bd, C, generator=g) * 0.5).to(dtype)
    src = torch.randperm(n_tx, generator=g)[:E] if route == "grouped" else torch.randint(0, n_tx, (E,), generator=g)
    dst = torch.randint(0, n_bd, (E,), generator=g)
    dst[dst == 7] = 8
    neg = (dst + torch.randint(1, n_bd, (E,), generator=g)) % n_bd
    a, b = z_tx.double().requires_grad_(True), z_bd.double().requires_grad_(True)
    ref = oracle.segmentation_loss(a, b, torch.stack([src, dst]), neg, "bce")
    (ref * 0.37).backward()
    da, db = z_tx.to(cuda).requires_grad_(True), z_bd.to(cuda).requires_grad_(True)
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), n_bd, n_tx, validate=False) if route == "grouped" else None
    loss = ops.bce_edge_loss(da, db, src.to(cuda), dst.to(cuda), neg.to(cuda), pos_groups=groups,
                             anchors_unique=route == "grouped")
    (loss * 0.37).backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    rtol, atol = (1e-5, 1e-7) if dtype == torch.float32 else (1e-2, 4e-6)
    assert torch.allclose(da.grad.cpu().double(), a.grad, rtol=rtol, atol=atol)
    assert torch.allclose(db.grad.cpu().double(), b.grad, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_loss_head_anchor_rows_equal_atomic_anchors(cuda, dtype, monkeypatch):
    """fp32 storage: loss_tx's anchor terms are STORED into a second gradient matrix (segger_triplet_args.grad_a_rows) and
    the l2-normalisation backward reads the sum of the two (segger_l2norm_bwd2) -- the loss head then differentiates
    through the un-normalised embeddings directly.  Same losses, same gradient of the un-normalised rows as the route
    with atomic anchors through ops.l2_normalize's own backward; masked rows (-1 positives) and a transcript that is
    nobody's positive / negative included.  bf16: the switch is a no-op by design (packed atomics kept)."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(21)
    n, nb, c, e = 3000, 40, 64, 1500
    y0 = torch.randn(n, c, generator=g).to(dtype)
    zb0 = torch.nn.functional.normalize(torch.randn(nb, c, generator=g), dim=-1).to(dtype)
    pos = torch.randint(0, n, (n,), generator=g); pos[::7] = -1
    neg = torch.randint(0, n, (n,), generator=g)
    bpos, bneg = torch.randint(0, nb, (nb,), generator=g), torch.randint(0, nb, (nb,), generator=g)
    dp, dn, w = torch.rand(nb, generator=g), torch.rand(nb, generator=g), torch.full((nb,), 1.0 / nb)
    src = torch.randperm(n, generator=g)[:e]
    dst = torch.randint(0, nb, (e,), generator=g)
    dneg = (dst + torch.randint(1, nb, (e,), generator=g)) % nb
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), nb, n, validate=False)
    a = torch.tensor([1.3, 1.0, 0.7], device=cuda)
    b = torch.tensor([0.5, 0.2, 0.3], device=cuda)
    out = {}
    monkeypatch.setattr(ops, "ONE_LAUNCH_LOSS_HEAD", False)       # (the round-3 routes; the one-launch head has its own test below)
    for rows in (True, False):
        monkeypatch.setattr(ops, "USE_ANCHOR_ROWS", rows)
        y = y0.to(cuda).requires_grad_(True)
        zb = zb0.to(cuda).requires_grad_(True)
        z = ops.l2_normalize(y)
        spec = ops.LossHeadSpec((torch.arange(n, device=cuda), pos.to(cuda), neg.to(cuda), 0.3, 1e-6),
                                (bpos.to(cuda), bneg.to(cuda), dp.to(cuda), dn.to(cuda), w.to(cuda), 1e-8),
                                (src.to(cuda), dst.to(cuda), dneg.to(cuda), 0.4, 1e-6, groups, True), tx_anchors_are_rows=True)
        res = ops.loss_head(z, zb, a, b, spec)
        (res[3] * 1.7 + res[0]).backward()
        out[rows] = (res.detach().clone(), y.grad.clone(), zb.grad.clone())
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-6, atol=1e-7)
    scale = out[False][1].float().abs().max().item()
    assert (out[True][1].float() - out[False][1].float()).abs().max().item() <= tol * scale + 1e-9
    assert torch.allclose(out[True][2].float(), out[False][2].float(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("unique", [True, False])
@pytest.mark.parametrize("prenorm", [True, False])
@pytest.mark.parametrize("kind", ["triplet", "bce"])
@pytest.mark.parametrize("dtype,C", [(torch.float32, 64), (torch.bfloat16, 64), (torch.float16, 32), (torch.bfloat16, 128)])
def test_one_launch_loss_head_equals_the_kernel_by_kernel_head(cuda, dtype, C, kind, prenorm, unique, monkeypatch):
    """segger_loss_head_fwd / _bwd (one launch each way: rows of the transcript gradient GATHERED over per-row contribution
    chains, stored once, no float atomics on that matrix, no zero fills, the row normalisation's backward inside) against
    the round-3 head (three loss kernels + combination, packed / fp32 atomics, separate l2norm backward): same four
    losses, same gradients of the (un-normalised or normalised) embeddings.  Masked rows (-1 positives), padded
    segmentation triplets (-1 positives), a transcript that is nobody's positive or negative, boundaries without
    triplets; ``unique=False``: one transcript anchors two segmentation triplets (the atomics fallback on top)."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(31 + C)
    n, nb, e = 3001, 45, 1400
    y0 = torch.randn(n, C, generator=g).to(dtype)
    yb0 = torch.randn(nb, C, generator=g).to(dtype)
    pos = torch.randint(0, n, (n,), generator=g); pos[::7] = -1
    neg = torch.randint(0, n, (n,), generator=g)
    pos[pos == 5] = 6; neg[neg == 5] = 6                               # row 5: in nobody's chain
    if dtype == torch.float32 or not prenorm:
        # (16-bit + prenorm: the normalised rows are rounded to the storage type before the loss reads them, and a hot row
        # sums hundreds of such roundings -- a property of the storage type, not of either head; checked in the other modes)
        pos[100:900:2] = 3; neg[1000:1300] = 9; pos[1500:1564] = 11; neg[1600:1665] = 12
    # rows 3 / 9 are HOT (400 / 300 draws: beyond what a chain holds, accumulated with atomics and finished by the last
    # contributor), row 11 fills its chain exactly (64), row 12 is hot by one
    bpos, bneg = torch.randint(0, nb, (nb,), generator=g), torch.randint(0, nb, (nb,), generator=g)
    dp, dn, w = torch.rand(nb, generator=g), torch.rand(nb, generator=g), torch.full((nb,), 1.0 / nb)
    w[3] = 0.0
    src = torch.randperm(n, generator=g)[:e]
    if not unique:
        src[7] = src[3]
    dst = torch.randint(0, nb - 3, (e,), generator=g)                  # the last boundaries have no triplet
    dst[-20:] = -1                                                     # padded triplets
    dneg = torch.where(dst >= 0, (dst + torch.randint(1, nb, (e,), generator=g)) % nb, torch.full_like(dst, -1))
    groups = csr_from_coo(dst.clamp(min=0).to(cuda), src.to(cuda), nb, n, validate=False)
    a = torch.tensor([1.3, 1.0, 0.7], device=cuda)
    b = torch.tensor([0.5, 0.2, 0.3], device=cuda)
    gvec = torch.tensor([1.0, 0.0, 0.5, 1.7], device=cuda)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "ONE_LAUNCH_LOSS_HEAD", fused)
        # the kernel-by-kernel head runs at fp32 storage on the SAME (already rounded) inputs: with 16-bit storage it rounds
        # every one of a hot row's hundreds of atomic adds to the storage type and is no reference for such rows
        y = (y0 if fused else y0.float()).to(cuda).requires_grad_(True)
        yb = (yb0 if fused else yb0.float()).to(cuda).requires_grad_(True)
        if prenorm:
            zs = ops.l2_normalize_many({"tx": y, "bd": yb})
            z, zb = zs["tx"], zs["bd"]
        else:
            z, zb = y, yb
        spec = ops.LossHeadSpec((torch.arange(n, device=cuda), pos.to(cuda), neg.to(cuda), 0.3, 1e-6),
                                (bpos.to(cuda), bneg.to(cuda), dp.to(cuda), dn.to(cuda), w.to(cuda), 1e-8),
                                (src.to(cuda), dst.to(cuda), dneg.to(cuda), 0.4, 1e-6, groups, unique), sg_kind=kind,
                                tx_anchors_are_rows=True)
        assert ops.loss_head_fused_supported(z, zb, spec) == fused
        res = ops.loss_head(z, zb, a, b, spec)
        res.backward(gvec)
        out[fused] = (res.detach().clone(), y.grad.float().clone(), yb.grad.float().clone())
    ltol = 2e-6 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert torch.allclose(out[True][0], out[False][0], rtol=ltol, atol=ltol * 1e-2)
    # fp32: two orders of the same sums; 16-bit: normalised rows and the stored gradient are rounded to the storage type once
    # (fp32 bound: a hot row adds up to 400 fp32 atomics in whatever order they arrive, in BOTH heads -- 400 x 2^-24 = 2.4e-5
    #  of the row's magnitude in the worst case; 2e-5 held in hundreds of runs and missed once by 5 %: 5e-5)
    tol = 5e-5 if dtype == torch.float32 else (3e-2 if dtype == torch.bfloat16 else 4e-3)
    for k in (1, 2):
        scale = out[False][k].abs().max().item()
        assert scale > 0 and (out[True][k] - out[False][k]).abs().max().item() <= tol * scale, (k, scale)


def test_one_launch_loss_head_run_to_run_spread(cuda):
    """No float atomic touches the transcript gradient any more: a row's terms are summed in fp32 registers and rounded
    ONCE.  What still varies between runs is the ORDER of a row's chain (the order in which the forward's atomicExch
    threaded the triplets): two runs agree to one rounding of the storage type (the round-3 head: one rounding per atomic
    add); the boundary side still adds atomically in fp32."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(77)
    n, nb, e, C = 20000, 300, 8000, 64
    y0 = torch.randn(n, C, generator=g).bfloat16()
    yb0 = torch.randn(nb, C, generator=g).bfloat16()
    pos, neg = torch.randint(0, n, (n,), generator=g), torch.randint(0, n, (n,), generator=g)
    bpos, bneg = torch.randint(0, nb, (nb,), generator=g), torch.randint(0, nb, (nb,), generator=g)
    dp, dn, w = torch.rand(nb, generator=g), torch.rand(nb, generator=g), torch.full((nb,), 1.0 / nb)
    src = torch.randperm(n, generator=g)[:e]
    dst = torch.randint(0, nb, (e,), generator=g)
    dneg = (dst + torch.randint(1, nb, (e,), generator=g)) % nb
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), nb, n, validate=False)
    a, b = torch.ones(3, device=cuda), torch.tensor([0.5, 0.2, 0.3], device=cuda)
    runs = []
    for _ in range(2):
        y, yb = y0.to(cuda).requires_grad_(True), yb0.to(cuda).requires_grad_(True)
        zs = ops.l2_normalize_many({"tx": y, "bd": yb})
        spec = ops.LossHeadSpec((torch.arange(n, device=cuda), pos.to(cuda), neg.to(cuda), 0.3, 1e-6),
                                (bpos.to(cuda), bneg.to(cuda), dp.to(cuda), dn.to(cuda), w.to(cuda), 1e-8),
                                (src.to(cuda), dst.to(cuda), dneg.to(cuda), 0.4, 1e-6, groups, True), tx_anchors_are_rows=True)
        ops.loss_head(zs["tx"], zs["bd"], a, b, spec)[3].backward()
        runs.append((y.grad.clone(), yb.grad.clone()))
    a0, a1 = runs[0][0].float(), runs[1][0].float()
    assert (a0 - a1).abs().max().item() <= 2.0 ** -7 * a0.abs().max().item()          # one bf16 ulp of the largest entry
    assert (a0 != a1).float().mean().item() < 0.05                                      # and most rows bit-identical
    assert torch.allclose(runs[0][1].float(), runs[1][1].float(), rtol=1e-2, atol=1e-6)


def test_masked_losses_equal_the_gathered_form(cuda):
    """loss_tx / loss_bd under a mask (no compaction, no host sync: the selector works under the mask, the fused
    triplet kernel skips the unmasked anchors) equal TripletLoss / MetricLoss on ``embeddings[mask], labels[mask]`` --
    the reference's call (lightning_model.py:158-165) -- for the same per-node uniform draws: same triplets, same
    values, same gradients."""
    from segger_amd.triplet_loss import MetricLoss, TripletLoss
    g = torch.Generator().manual_seed(9)
    n, c, k = 5000, 64, 6
    a = torch.randn(k, 4, generator=g); a = a / a.norm(dim=1, keepdim=True)
    sim = a @ a.t()
    z = torch.nn.functional.normalize(torch.randn(n, c, generator=g), dim=-1)
    labels = torch.randint(0, k, (n,), generator=g)
    labels[labels == 2] = 3                                   # an absent cluster
    mask = (torch.rand(n, generator=g) < 0.8).to(cuda)
    u = tuple(torch.rand(n, generator=g).to(cuda) for _ in range(4))
    um = tuple(t[mask] for t in u)
    lab = labels.to(cuda)
    idx = mask.nonzero().squeeze(1)
    loss_fn = TripletLoss(sim, margin=0.3)
    # the triplets themselves: masked selection == selection on the compacted arrays, mapped back
    pos_m, neg_m, dp_m, dn_m = loss_fn.selector.sample_triplets(lab, u, mask=mask)
    pos_c, neg_c, dp_c, dn_c = loss_fn.selector.sample_triplets(lab[mask], um)
    assert torch.equal(pos_m[mask], idx[pos_c]) and torch.equal(neg_m[mask], idx[neg_c])
    assert bool((pos_m[~mask] == -1).all()) and torch.equal(dp_m[mask], dp_c) and torch.equal(dn_m[mask], dn_c)
    zd = z.to(cuda).requires_grad_(True)
    l1 = loss_fn.forward_masked(zd, lab, mask, uniforms=u)
    (l1 * 0.5).backward()
    zr = z.to(cuda).requires_grad_(True)
    l2 = loss_fn.forward(zr[mask], lab[mask], uniforms=um)
    (l2 * 0.5).backward()
    assert abs(l1.item() - l2.item()) < 1e-6
    assert torch.allclose(zd.grad, zr.grad, atol=1e-7, rtol=1e-4)
    ml = MetricLoss(sim)
    z1, z2 = z.to(cuda).requires_grad_(True), z.to(cuda).requires_grad_(True)
    m1 = ml.forward_masked(z1, lab, mask, uniforms=u)
    m2 = ml.forward(z2[mask], lab[mask], uniforms=um)
    m1.backward(); m2.backward()
    assert abs(m1.item() - m2.item()) < 1e-6 and torch.allclose(z1.grad, z2.grad, atol=1e-7, rtol=1e-4)
    # an empty mask: zero loss, zero gradient, no NaN
    none = torch.zeros_like(mask)
    z3 = z.to(cuda).requires_grad_(True)
    l3 = loss_fn.forward_masked(z3, lab, none) + ml.forward_masked(z3, lab, none)
    l3.backward()
    assert l3.item() == 0.0 and not z3.grad.any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_posfreq_matches_reference_formula(oracle, cuda, dtype):
    from segger_amd import ops
    g = torch.Generator().manual_seed(3)
    n = 3000
    pos = torch.rand(n, 2, generator=g) * 500 + 10
    batch = torch.sort(torch.randint(0, 5, (n,), generator=g)).values
    mins, maxs = ops.segment_minmax(pos.to(cuda), batch.to(cuda), 6)
    for b in range(5):
        m = batch == b
        assert torch.equal(mins[b].cpu(), pos[m].min(0).values) and torch.equal(maxs[b].cpu(), pos[m].max(0).values)
    assert (mins[5] == 0).all() and (maxs[5] == 0).all()                       # graph without nodes
    # one-launch form: buffers armed by the caller, empty graphs left at (+inf, -inf)
    inf = float("inf")
    armed = (torch.full((6, 2), inf, device=cuda), torch.full((6, 2), -inf, device=cuda))
    m2, x2 = ops.segment_minmax(pos.to(cuda), batch.to(cuda), 6, keep_empty=True, out=armed)
    assert m2 is armed[0] and torch.equal(m2[:5], mins[:5]) and torch.equal(x2[:5], maxs[:5])
    assert (m2[5] == inf).all() and (x2[5] == -inf).all()
    out = ops.posfreq(pos.to(cuda), batch.to(cuda), mins, maxs, 256, dtype)
    p = oracle.normalize_positions(pos.double(), batch)
    ref = oracle.sinusoidal_embedding(p.flatten(), 256, 10000).reshape(n, 2, 256)
    tol = 2e-6 if dtype == torch.float32 else 4e-3
    assert (out.double().cpu() - ref).abs().max() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_gelu_and_l2norm_autograd(cuda, dtype):
    from segger_amd import ops
    g = torch.Generator().manual_seed(4)
    n, G, D = 5000, 37, 64
    table = torch.randn(G, D, generator=g)
    ids = torch.randint(0, G - 2, (n,), generator=g)                           # two genes never occur
    pe = torch.randn(n, D, generator=g).to(dtype)
    gx = torch.randn(n, 2 * D, generator=g).to(dtype)
    t_d, pe_d = table.to(cuda).requires_grad_(True), pe.to(cuda).requires_grad_(True)
    x0 = ops.embed_gelu(t_d, ids.to(cuda), pe_d)
    x0.backward(gx.to(cuda))
    t_r, pe_r = table.double().requires_grad_(True), pe.double().requires_grad_(True)
    ref = torch.nn.functional.gelu(torch.cat((t_r[ids], pe_r), -1))
    ref.backward(gx.double())
    rt = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.allclose(x0.double().cpu(), ref, rtol=rt, atol=rt)
    assert torch.allclose(pe_d.grad.double().cpu(), pe_r.grad, rtol=rt, atol=rt)
    assert torch.allclose(t_d.grad.double().cpu(), t_r.grad, rtol=1e-4, atol=1e-3 if dtype == torch.float32 else 5e-2)
    assert (t_d.grad[G - 2:] == 0).all()
    # L2 normalisation
    y = torch.randn(n, 64, generator=g).to(dtype); y[7] = 0                      # a zero row stays zero
    gz = torch.randn(n, 64, generator=g).to(dtype)
    y_d = y.to(cuda).requires_grad_(True)
    z = ops.l2_normalize(y_d)
    z.backward(gz.to(cuda))
    y_r = y.double().requires_grad_(True)
    z_r = torch.nn.functional.normalize(y_r, dim=-1)
    z_r.backward(gz.double())
    assert torch.allclose(z.double().cpu(), z_r, rtol=rt, atol=rt)
    ok = torch.ones(n, dtype=torch.bool); ok[7] = False                          # torch: grad at the clamp is gz/eps
    assert torch.allclose(y_d.grad.double().cpu()[ok], y_r.grad[ok], rtol=rt, atol=rt * 3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_table_grad_skewed_ids(cuda, dtype):
    """One gene owns 60 % of the rows (-> 32 chunks of ~1300 rows), many genes are empty or tiny; the cached
    rows-by-gene view and the view built inside backward give the same, deterministic gradient."""
    from segger_amd import ops
    g = torch.Generator().manual_seed(9)
    n, G, D = 70_001, 300, 128
    ids = torch.randint(0, 150, (n,), generator=g)
    ids[torch.rand(n, generator=g) < 0.6] = 5
    table = torch.randn(G, D, generator=g)
    pe = torch.randn(n, D, generator=g).to(dtype)
    gx = torch.randn(n, 2 * D, generator=g).to(dtype)
    grads = []
    for cached in (True, False, True):
        t_d, pe_d = table.to(cuda).requires_grad_(True), pe.to(cuda).requires_grad_(True)
        ids_d = ids.to(cuda).int()
        by_gene = ops.rows_by_id(ids_d, G) if cached else None
        ops.embed_gelu(t_d, ids_d, pe_d, by_gene).backward(gx.to(cuda))
        grads.append(t_d.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    ref = torch.zeros(G, D, dtype=torch.float64).index_add_(0, ids, gx[:, :D].double())
    x = table.double()
    cdf = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    ref = ref * (cdf + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5)
    err = (grads[0].double().cpu() - ref).abs()
    bound = 2e-6 * torch.zeros(G, D, dtype=torch.float64).index_add_(0, ids, gx[:, :D].double().abs()) + 1e-5
    assert bool((err <= bound).all()), float((err / bound).max())
    assert (grads[0][150:] == 0).all()


def test_gelu_kernel_accuracy(cuda):
    """The erf-free normal CDF behind every fused GELU: |gelu - exact| <= 5e-7 on [-8, 8] (fp32 path; torch's
    own fp32 gelu is within 1.2e-6 of the exact value on the same range)."""
    from segger_amd import ops
    n, D = 4096, 32
    x = torch.linspace(-8, 8, n * D, dtype=torch.float64).reshape(n, D)
    table = torch.zeros(1, D, device=cuda)
    out = ops.embed_gelu(table, torch.zeros(n, dtype=torch.int32, device=cuda), x.float().to(cuda))[:, D:]
    xs = x.float().double()
    ref = xs * 0.5 * torch.erfc(-xs / 2 ** 0.5)
    err = (out.double().cpu() - ref).abs()
    assert float(err.max()) <= 5e-7


def test_sampler_stream_is_fresh_per_call_and_follows_manual_seed(cuda):
    """Without given uniforms the fused sampler draws from its own counter-based stream: seeded from torch's CPU
    generator (eager) or from a device word (graph replays) -- never the same draws twice in a row; the drawn
    positives share the anchor's cluster at the rate the similarity matrix dictates."""
    from segger_amd.triplet_loss import FastTripletSelector
    k, n = 6, 20000
    g = torch.Generator().manual_seed(3)
    sim = torch.rand(k, k, generator=g) * 2 - 1
    sim = (sim + sim.t()) / 2
    lab = torch.randint(0, k, (n,), generator=g).to(cuda)
    sel = FastTripletSelector(sim)
    ix = sel.build_index(lab)
    torch.manual_seed(11)
    a = sel.sample_triplets(lab, index=ix)
    b = sel.sample_triplets(lab, index=ix)
    torch.manual_seed(11)
    c = sel.sample_triplets(lab, index=ix)
    assert not torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a, c))
    word = torch.zeros(1, dtype=torch.int64, device=cuda)
    d0 = sel.sample_triplets(lab, index=ix, device_seed=(5, word))
    d0b = sel.sample_triplets(lab, index=ix, device_seed=(5, word))
    word += 256
    d1 = sel.sample_triplets(lab, index=ix, device_seed=(5, word))
    assert torch.equal(d0[0], d0b[0]) and not torch.equal(d0[0], d1[0])
    # P(positive in own cluster) = sim'[r, r] / sum_c sim'[r, c] over present clusters, sim' = clamp(sim with diag 1)
    s = sel.similarity
    want = float((s.diag() / s.sum(1))[lab.cpu()].mean())
    got = float((lab[a[0]] == lab).float().mean())
    assert abs(got - want) < 0.02


def test_sample_negatives_is_a_uniform_nonzero_shift(cuda):
    """segger_sample_negatives == (pos + randint(1, n_b)) % n_b (lightning_model.py:178-180): never the positive
    itself, every other row equally likely; -1 positives stay -1; n_b may come from the device."""
    from segger_amd import ops
    n_b, n = 7, 70000
    pos = torch.randint(0, n_b, (n,), device=cuda)
    pos[::10] = -1
    torch.manual_seed(2)
    neg = ops.sample_negatives(pos, n_b)
    live = pos >= 0
    assert bool((neg[~live] == -1).all())
    assert bool(((neg[live] >= 0) & (neg[live] < n_b) & (neg[live] != pos[live])).all())
    shift = (neg[live] - pos[live]) % n_b
    freq = torch.bincount(shift, minlength=n_b).float() / live.sum()
    assert freq[0] == 0 and bool(((freq[1:] - 1 / (n_b - 1)).abs() < 0.01).all())
    assert not torch.equal(neg, ops.sample_negatives(pos, n_b))           # a fresh draw per call
    torch.manual_seed(2)
    assert torch.equal(neg, ops.sample_negatives(pos, n_b))
    dev_n = torch.tensor([n_b], device=cuda)
    torch.manual_seed(2)
    assert torch.equal(neg, ops.sample_negatives(pos, 0, dev_n))
    assert bool((ops.sample_negatives(pos.clamp(min=0) * 0, 1) == 0).all())   # a single boundary: the reference skips


def test_stage_segments_copy_and_fill(cuda):
    """segger_stage: every fill rule against its torch formula, mixed element sizes, one launch."""
    from segger_amd import ops
    src64 = torch.arange(100, 137, device=cuda)
    srcf = torch.randn(11, 2, device=cuda)
    srcb = torch.tensor([True, False, True], device=cuda)
    d_narrow = torch.full((50,), -7, dtype=torch.int32, device=cuda)
    d_tile = torch.zeros(20, 2, device=cuda)
    d_bool = torch.ones(9, dtype=torch.bool, device=cuda)
    d_div = torch.zeros(30, dtype=torch.int32, device=cuda)
    d_mod = torch.zeros(30, dtype=torch.int64, device=cuda)
    d_ramp = torch.zeros(12, dtype=torch.int64, device=cuda)
    d_const = torch.zeros(3, device=cuda)
    d_full = torch.zeros(37, dtype=torch.int64, device=cuda)
    ops.stage([(d_narrow, src64, "const", 5, 0, 0), (d_tile, srcf, "tile", 2, 0, 0), (d_bool, srcb, "const", 0, 0, 0),
               (d_div, src64[:4].to(torch.int32), "div", 1000, 3, 0), (d_mod, src64[:4], "mod", 50, 7, 0),
               (d_ramp, src64[:5], "ramp", 200, 4, 10), (d_const, None, "const", ops.float_bits(2.5), 0, 0),
               (d_full, src64, "const", 0, 0, 0)], cuda)
    k = torch.arange(64, device=cuda)
    assert torch.equal(d_narrow, torch.cat([src64.int(), torch.full((13,), 5, dtype=torch.int32, device=cuda)]))
    assert torch.equal(d_tile, torch.cat([srcf, srcf[:1].expand(9, 2)]))
    assert torch.equal(d_bool, torch.tensor([1, 0, 1, 0, 0, 0, 0, 0, 0], dtype=torch.bool, device=cuda))
    assert torch.equal(d_div, torch.cat([src64[:4].int(), (1000 + k[:26] // 3).int()]))
    assert torch.equal(d_mod, torch.cat([src64[:4], 50 + k[:26] % 7]))
    assert torch.equal(d_ramp, torch.cat([src64[:5], 200 + torch.clamp((k[:7] + 1) * 4, max=10)]))
    assert torch.equal(d_const, torch.full((3,), 2.5, device=cuda))
    assert torch.equal(d_full, src64)
    with pytest.raises(TypeError):
        ops.stage([(d_const, src64[:2], "const", 0, 0, 0)], cuda)
    many = [(torch.zeros(5, dtype=torch.int64, device=cuda), None, "div", i, 1, 0) for i in range(101)]   # > one launch
    ops.stage(many, cuda)
    assert all(torch.equal(t[0], i + k[:5]) for i, t in enumerate(many))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n", [1, 31, 1000, 40_003])
def test_fused_positional_embedder_matches_unfused_route(cuda, dtype, n):
    """segger_posmlp_fwd (sinusoid generated as MFMA fragments, both Linear layers and SiLU fused) against fp64
    arithmetic of ist_encoder.py:33-79 on the same rounded weights, and its gradients against autograd of the unfused
    route (posfreq + linear + SiLU + linear).  Tolerances: one rounding of a 64-term dot product of O(1) values to the
    16-bit output (2^-8 / 2^-10 relative + the rounding of h1)."""
    from segger_amd import ops
    from segger_amd.ist_encoder import Positional2dEmbedder
    g = torch.Generator().manual_seed(n)
    pos = (torch.rand(n, 2, generator=g) * 700 + 3).to(cuda)
    batch = torch.sort(torch.randint(0, 3, (n,), generator=g)).values.to(cuda)
    emb = Positional2dEmbedder(128).to(cuda)
    with torch.no_grad():
        for p_ in emb.parameters():
            p_.mul_(3.0)                                     # bigger activations than the default init gives
    pe = emb(pos, batch, num_graphs=3, dtype=dtype)          # fused (CUDA, 16-bit, 256 -> 64 -> 64)
    assert pe.shape == (n, 128) and pe.dtype == dtype
    # fp64 reference on the weights as the kernel sees them (rounded to dtype)
    mins, maxs = ops.segment_minmax(pos, batch, 3)
    p = ((pos - mins[batch]) / (maxs[batch] - mins[batch] + 1e-8)).double().flatten()
    j = torch.arange(128, device=cuda, dtype=torch.float64)
    ang = p[:, None] * torch.exp(-torch.log(torch.tensor(10000.0, dtype=torch.float64)) * j / 128)[None]
    feat = torch.cat([ang.cos(), ang.sin()], 1).to(dtype).double()
    l0, l2 = emb.mlp[0], emb.mlp[2]
    z1 = feat @ l0.weight.detach().to(dtype).double().t() + l0.bias.detach().double()
    h1 = (z1 * torch.sigmoid(z1)).to(dtype).double()
    ref = (h1 @ l2.weight.detach().to(dtype).double().t() + l2.bias.detach().double()).reshape(n, 128)
    rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert bool(((pe.double() - ref).abs() <= 2 * rel * ref.abs() + 4 * rel).all())
    # gradients: fused forward + assembled backward vs autograd through the unfused ops
    gy = torch.randn(n, 128, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1)).to(dtype)
    emb.zero_grad()
    emb(pos, batch, num_graphs=3, dtype=dtype).backward(gy)
    got = [p_.grad.clone() for p_ in emb.parameters()]
    emb.zero_grad()
    freq = ops.posfreq(pos, batch, mins, maxs, 256, dtype)
    h = torch.nn.functional.silu(ops.linear(freq, l0.weight, l0.bias))
    ops.linear(h, l2.weight, l2.bias).flatten(-2).backward(gy)
    for a, p_ in zip(got, emb.parameters()):
        scale = p_.grad.abs().max().item() + 1e-6
        assert (a - p_.grad).abs().max().item() <= 3e-2 * scale
    # the one-pass backward (segger_posmlp_bwd) against round 2's three kernels and against fp64 on the stored z1
    ops.FUSED_POSMLP_BWD = False
    try:
        emb.zero_grad()
        emb(pos, batch, num_graphs=3, dtype=dtype).backward(gy)
    finally:
        ops.FUSED_POSMLP_BWD = True
    g64 = gy.double().reshape(-1, 64)
    z1r = z1.to(dtype).double()
    sg = torch.sigmoid(z1r)
    dz1 = ((g64 @ l2.weight.detach().to(dtype).double()) * (sg * (1 + z1r * (1 - sg)))).to(dtype).double()
    ref64 = [dz1.t() @ feat, dz1.sum(0), g64.t() @ (z1r * sg).to(dtype).double(), g64.sum(0)]
    for a, p_, r64 in zip(got, emb.parameters(), ref64):
        scale = r64.abs().max().item() + 1e-6
        assert (a - p_.grad).abs().max().item() <= 1e-2 * scale
        assert (a.double() - r64).abs().max().item() <= 1e-2 * scale
    with torch.no_grad():                                    # inference: same values, nothing stored
        assert torch.equal(emb(pos, batch, num_graphs=3, dtype=dtype), pe)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("na,nb", [(40_003, 517), (1000, 1), (7, 5000), (300, 0), (0, 41)])
def test_positional_embedder_pair_is_one_node_with_the_summed_gradients(cuda, dtype, na, nb):
    """ops.posmlp_pair (two row sets, ONE autograd node, ONE segger_posmlp_bwd_pair launch) against two ops.posmlp nodes:
    outputs bit-identical (same forward kernel), parameter gradients equal to the sum autograd forms from the two nodes up to
    the order of the fp32 partial sums; one side without a gradient; the encoder's large-batch route takes it."""
    from segger_amd import ops
    from segger_amd.ist_encoder import Positional2dEmbedder
    g = torch.Generator().manual_seed(na + nb)
    pos_a = (torch.rand(na, 2, generator=g) * 500).to(cuda)
    pos_b = (torch.rand(nb, 2, generator=g) * 500).to(cuda)
    ba = torch.sort(torch.randint(0, 3, (na,), generator=g)).values.to(cuda)
    bb = torch.sort(torch.randint(0, 3, (nb,), generator=g)).values.to(cuda)
    emb = Positional2dEmbedder(128).to(cuda)
    with torch.no_grad():
        for p_ in emb.parameters():
            p_.mul_(3.0)
    l0, l2 = emb.mlp[0], emb.mlp[2]
    mm_a, mm_b = ops.segment_minmax(pos_a, ba, 3, keep_empty=True), ops.segment_minmax(pos_b, bb, 3, keep_empty=True)
    args = (l0.weight, l0.bias, l2.weight, l2.bias, dtype)
    gen = torch.Generator(device=cuda).manual_seed(2)
    g_a = torch.randn(na, 128, device=cuda, generator=gen).to(dtype)
    g_b = torch.randn(nb, 128, device=cuda, generator=gen).to(dtype)

    def separate(use_a=True, use_b=True):
        emb.zero_grad()
        act, pre = ops.posmlp(pos_a, ba, mm_a[0], mm_a[1], *args, gelu=True, return_pre=True)
        pe_b = ops.posmlp(pos_b, bb, mm_b[0], mm_b[1], *args)
        torch.autograd.backward([t for t, u in ((pre, use_a), (pe_b, use_b)) if u], [t for t, u in ((g_a, use_a), (g_b, use_b)) if u])
        return act, pre.detach(), pe_b.detach(), [p_.grad.clone() for p_ in emb.parameters()]

    def pair(use_a=True, use_b=True):
        emb.zero_grad()
        (act, pre), pe_b = ops.posmlp_pair(pos_a, ba, mm_a[0], mm_a[1], pos_b, bb, mm_b[0], mm_b[1], *args)
        assert not act.requires_grad and pre.grad_fn is pe_b.grad_fn          # one node
        torch.autograd.backward([t for t, u in ((pre, use_a), (pe_b, use_b)) if u], [t for t, u in ((g_a, use_a), (g_b, use_b)) if u])
        return act, pre.detach(), pe_b.detach(), [p_.grad.clone() for p_ in emb.parameters()]

    for use in ((True, True), (True, False), (False, True)):
        ref, got = separate(*use), pair(*use)
        for a, b in zip(ref[:3], got[:3]):
            assert torch.equal(a, b)
        for a, b in zip(ref[3], got[3]):
            assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item() + 1e-7
    # the encoder: one call per node type on the split route -> one node when the switch is on, same step either way
    import segger_amd.ist_encoder as ie
    assert ie._pair_node(emb, pos_a, ba, pos_b, bb, 3, dtype) is not None
    assert ie._pair_node(emb, pos_a, ba, pos_b, bb, 3, torch.float32) is None          # (the fused embedder is 16-bit)
    with torch.no_grad():
        assert ie._pair_node(emb, pos_a, ba, pos_b, bb, 3, dtype) is None              # (nothing to differentiate)


def test_step_draws_equal_the_separate_launches(cuda):
    """segger_step_draws (all random draws of a training step in one launch: bit planes of the edge views, both triplet
    samplers, the negative boundaries, + the increment of Adam's step counters) against dropout_bits_many, triplet_sample x 2
    and sample_negatives with the same seeds and the same device seed word: bit-identical."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    from segger_amd.triplet_loss import FastTripletSelector
    g = torch.Generator().manual_seed(5)
    n, nb, e = 5003, 211, 40_007
    src, dst = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
    by_dst = csr_from_coo(dst.to(cuda), src.to(cuda), n, n, validate=False)
    by_src = csr_from_coo(src.to(cuda), dst.to(cuda), n, n, validate=False)
    e2 = 1234
    tb = csr_from_coo(torch.randint(0, nb, (e2,), generator=g).to(cuda), torch.randint(0, n, (e2,), generator=g).to(cuda), nb, n,
                      validate=False)
    views = [(by_dst, [0, 2, 4, 6, 8]), (by_src, [0, 2, 4, 6, 8]), (tb, [1, 3, 5, 7, 9])]
    k = 7
    sim = torch.rand(k, k, generator=g); sim = (sim + sim.T) / 2
    sel = FastTripletSelector(sim.to(cuda))
    lab_tx = torch.randint(0, k, (n,), generator=g).to(cuda)
    lab_bd = torch.randint(0, k, (nb,), generator=g).to(cuda)
    ix_tx = sel.build_index(lab_tx, mask=(torch.rand(n, generator=g) > 0.1).to(cuda))
    ix_bd = sel.build_index(lab_bd, mask=None)
    sg_pos = torch.randint(0, nb, (e2,), generator=g).to(cuda); sg_pos[-10:] = -1
    n_bd_dev = torch.tensor([nb], device=cuda)
    word = torch.tensor([256 * 17], dtype=torch.int64, device=cuda)
    counters = [torch.full((), float(i), device=cuda) for i in range(5)]
    planes, samples, negs = ops.step_draws(views, 2, 0.2, [(ix_tx, 0x7478), (ix_bd, 0x6264)], (sg_pos, 0, n_bd_dev, 0x7367), word,
                                           advance=counters)
    ref_planes = ops.dropout_bits_many(views, 2, 0.2, word)
    for a, b in zip(planes, ref_planes):
        assert torch.equal(a, b)
    for (ix, seed), got in zip(((ix_tx, 0x7478), (ix_bd, 0x6264)), samples):
        ref = ops.triplet_sample(ix, seed=seed, seed_dev=word)
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    assert torch.equal(negs, ops.sample_negatives(sg_pos, 0, n_bd_dev, seed=0x7367, seed_dev=word))
    assert [float(c) for c in counters] == [1.0, 2.0, 3.0, 4.0, 5.0]
    # nothing but samplers / nothing but planes
    _, s2, n2 = ops.step_draws([], 2, 0.0, [(ix_bd, 3)], None, word)
    assert n2 is None and torch.equal(s2[0][0], ops.triplet_sample(ix_bd, seed=3, seed_dev=word)[0])
    p3, s3, _ = ops.step_draws(views[:1], 2, 0.5, [], None, word)
    assert s3 == [] and torch.equal(p3[0], ops.dropout_bits_many(views[:1], 2, 0.5, word)[0])


@pytest.mark.parametrize("kind", ["triplet", "bce"])
def test_one_launch_loss_head_deferred_finish(cuda, kind):
    """``LossHeadSpec.defer_finish`` (a captured training step's form): the forward leaves its per-block partial sums, the
    backward launch -- one extra workgroup -- finishes the three losses into the forward's output and takes its scale
    factors from the hinted gradient directly: same losses (once the backward has run) and same gradients as the form with
    the finishing launch."""
    from segger_amd import ops
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(3)
    n, nb, e, C = 7001, 90, 2500, 64
    y0 = torch.randn(n, C, generator=g).bfloat16()
    yb0 = torch.randn(nb, C, generator=g).bfloat16()
    pos, neg = torch.randint(0, n, (n,), generator=g), torch.randint(0, n, (n,), generator=g)
    pos[::9] = -1
    bpos, bneg = torch.randint(0, nb, (nb,), generator=g), torch.randint(0, nb, (nb,), generator=g)
    dp, dn, w = torch.rand(nb, generator=g), torch.rand(nb, generator=g), torch.full((nb,), 1.0 / nb)
    src = torch.randperm(n, generator=g)[:e]
    dst = torch.randint(0, nb, (e,), generator=g)
    dneg = (dst + torch.randint(1, nb, (e,), generator=g)) % nb
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), nb, n, validate=False)
    a, b = torch.tensor([1.2, 1.0, 0.8], device=cuda), torch.tensor([0.5, 0.2, 0.3], device=cuda)
    hint = torch.tensor([0.0, 0.0, 0.0, 1.0], device=cuda)
    out = {}
    for defer in (False, True):
        y, yb = y0.to(cuda).requires_grad_(True), yb0.to(cuda).requires_grad_(True)
        zs = ops.l2_normalize_many({"tx": y, "bd": yb})
        spec = ops.LossHeadSpec((torch.arange(n, device=cuda), pos.to(cuda), neg.to(cuda), 0.3, 1e-6),
                                (bpos.to(cuda), bneg.to(cuda), dp.to(cuda), dn.to(cuda), w.to(cuda), 1e-8),
                                (src.to(cuda), dst.to(cuda), dneg.to(cuda), 0.4, 1e-6, groups, True), sg_kind=kind,
                                tx_anchors_are_rows=True, grad_out_hint=hint)
        spec.defer_finish = defer
        res = ops.loss_head(zs["tx"], zs["bd"], a, b, spec)
        res.backward(hint)
        out[defer] = (res.detach().clone(), y.grad.float().clone(), yb.grad.float().clone())
    assert torch.equal(out[True][0], out[False][0])                   # same partial sums, same fixed-order finish
    assert out[False][0][3] > 0
    for k in (1, 2):
        scale = out[False][k].abs().max().item()
        assert (out[True][k] - out[False][k]).abs().max().item() <= 2e-2 * scale


def _hot_row_problem(cuda, dtype=torch.float32, C=64, seed=91):
    from segger_amd.graph import csr_from_coo
    g = torch.Generator().manual_seed(seed)
    n, nb, e = 3001, 45, 1400
    y0 = torch.randn(n, C, generator=g).to(dtype)
    yb0 = torch.randn(nb, C, generator=g).to(dtype)
    pos = torch.randint(0, n, (n,), generator=g); pos[::7] = -1
    neg = torch.randint(0, n, (n,), generator=g)
    pos[100:900:2] = 3; neg[1000:1300] = 9; pos[1500:1564] = 11; neg[1600:1665] = 12      # rows 3, 9, 12 hot; 11 exactly full
    bpos, bneg = torch.randint(0, nb, (nb,), generator=g), torch.randint(0, nb, (nb,), generator=g)
    dp, dn, w = torch.rand(nb, generator=g), torch.rand(nb, generator=g), torch.full((nb,), 1.0 / nb)
    src = torch.randperm(n, generator=g)[:e]
    dst = torch.randint(0, nb, (e,), generator=g)
    dneg = (dst + torch.randint(1, nb, (e,), generator=g)) % nb
    groups = csr_from_coo(dst.to(cuda), src.to(cuda), nb, n, validate=False)

    def spec():
        from segger_amd import ops
        return ops.LossHeadSpec((torch.arange(n, device=cuda), pos.to(cuda), neg.to(cuda), 0.3, 1e-6),
                                (bpos.to(cuda), bneg.to(cuda), dp.to(cuda), dn.to(cuda), w.to(cuda), 1e-8),
                                (src.to(cuda), dst.to(cuda), dneg.to(cuda), 0.4, 1e-6, groups, True), tx_anchors_are_rows=True)
    return y0, yb0, spec


@pytest.mark.parametrize("prenorm", [True, False])
@pytest.mark.parametrize("how", ["b0", "a0", "gout"])
def test_one_launch_loss_head_hot_rows_with_zero_tx_weight(cuda, how, prenorm, monkeypatch):
    """A hot row waits for as many arrivals as the FORWARD flagged (from the unscaled weights).  With the transcript loss
    weighted by zero (``tx_weight_start=0``: b[0] = 0; or a[0] = 0; or a zero incoming gradient) every scaled contribution
    is zero -- the flagged ones must still arrive, or the row is never finished and its slot of the (torch.empty) gradient
    matrix stays uninitialised.  The gradient buffer is poisoned with NaN through the caching allocator first."""
    from segger_amd import ops
    y0, yb0, mk = _hot_row_problem(cuda)
    a = torch.tensor([0.0 if how == "a0" else 1.3, 1.0, 0.7], device=cuda)
    b = torch.tensor([0.0 if how == "b0" else 0.5, 0.2, 0.3], device=cuda)
    gvec = torch.tensor([0.0, 0.3, 0.5, 0.0 if how == "gout" else 1.7], device=cuda)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "ONE_LAUNCH_LOSS_HEAD", fused)
        y, yb = y0.to(cuda).requires_grad_(True), yb0.to(cuda).requires_grad_(True)
        if prenorm:
            zs = ops.l2_normalize_many({"tx": y, "bd": yb})
            z, zb = zs["tx"], zs["bd"]
        else:
            z, zb = y, yb
        spec = mk()
        assert ops.loss_head_fused_supported(z, zb, spec) == fused
        res = ops.loss_head(z, zb, a, b, spec)
        poison = torch.full((y0.shape[0], y0.shape[1]), float("nan"), device=cuda)      # what torch.empty hands out next
        del poison
        res.backward(gvec)
        out[fused] = (res.detach().clone(), y.grad.clone(), yb.grad.clone())
    assert torch.isfinite(out[True][1]).all() and torch.isfinite(out[True][2]).all()
    assert torch.allclose(out[True][0], out[False][0], rtol=2e-6, atol=1e-8)
    for k in (1, 2):
        scale = out[False][k].abs().max().item()
        assert (out[True][k] - out[False][k]).abs().max().item() <= 2e-5 * scale + 1e-12, (k, scale)
    if how != "gout":
        assert out[True][1][[3, 9, 12]].abs().max().item() >= 0.0          # hot rows present and finite (checked above)


def test_one_launch_loss_head_backward_twice(cuda):
    """retain_graph=True and a second backward: the boundary gradient and the hot rows are re-armed, so the second pass
    returns what the first did (not twice the boundary gradient, not stale hot rows)."""
    from segger_amd import ops
    y0, yb0, mk = _hot_row_problem(cuda, seed=93)
    a = torch.tensor([1.3, 1.0, 0.7], device=cuda)
    b = torch.tensor([0.5, 0.2, 0.3], device=cuda)
    y, yb = y0.to(cuda).requires_grad_(True), yb0.to(cuda).requires_grad_(True)
    zs = ops.l2_normalize_many({"tx": y, "bd": yb})
    spec = mk()
    assert ops.loss_head_fused_supported(zs["tx"], zs["bd"], spec)
    res = ops.loss_head(zs["tx"], zs["bd"], a, b, spec)
    g1 = torch.autograd.grad(res[3], (y, yb), retain_graph=True)
    poison = torch.full_like(y0.to(cuda), float("nan")); del poison
    g2 = torch.autograd.grad(res[3], (y, yb))
    for u, v in zip(g1, g2):
        assert torch.isfinite(v).all()
        assert (u - v).abs().max().item() <= 2e-5 * u.abs().max().item()


@pytest.mark.parametrize("n_bd", [0, 1, 411])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_front_join_autograd(cuda, dtype, n_bd):
    """segger_front_join_fwd / _bwd: (gelu(cat(E[g], pe[:n_tx])), gelu(cat(x_bd, pe[n_tx:]))) of both node types in one
    launch each way (reference ist_encoder.py:312-320 per type) against torch in float64: outputs, the gradient of the
    un-split positional matrix, of the boundary features and of the embedding table (genes that never occur: zero rows).
    Inputs with a row stride (views into wider matrices) included."""
    from segger_amd import ops
    g = torch.Generator().manual_seed(40 + n_bd)
    n, G, D = 3000, 37, 128
    table = torch.randn(G, D, generator=g)
    ids = torch.randint(0, G - 2, (n,), generator=g)
    pe = torch.randn(n + n_bd, D, generator=g).to(dtype)
    xb_wide = torch.randn(n_bd, D + 64, generator=g).to(dtype)                 # boundary features as a strided view
    g_tx = torch.randn(n, 2 * D, generator=g).to(dtype)
    g_bd = torch.randn(n_bd, 2 * D, generator=g).to(dtype)
    t_d, pe_d = table.to(cuda).requires_grad_(True), pe.to(cuda).requires_grad_(True)
    xbw_d = xb_wide.to(cuda).requires_grad_(True)
    by_gene = ops.rows_by_id(ids.to(cuda).to(torch.int32), G)
    x_tx, x_bd = ops.front_join(t_d, ids.to(cuda), xbw_d[:, :D], pe_d, by_gene)
    assert x_tx.shape == (n, 2 * D) and x_bd.shape == (n_bd, 2 * D)
    torch.autograd.backward((x_tx, x_bd), (g_tx.to(cuda), g_bd.to(cuda)))
    t_r, pe_r, xb_r = table.double().requires_grad_(True), pe.double().requires_grad_(True), xb_wide.double().requires_grad_(True)
    r_tx = torch.nn.functional.gelu(torch.cat((t_r[ids], pe_r[:n]), -1))
    r_bd = torch.nn.functional.gelu(torch.cat((xb_r[:, :D], pe_r[n:]), -1))
    torch.autograd.backward((r_tx, r_bd), (g_tx.double(), g_bd.double()))
    rt = 1e-5 if dtype == torch.float32 else (1e-2 if dtype == torch.bfloat16 else 2e-3)
    assert torch.allclose(x_tx.double().cpu(), r_tx, rtol=rt, atol=rt)
    assert torch.allclose(x_bd.double().cpu(), r_bd, rtol=rt, atol=rt)
    assert torch.allclose(pe_d.grad.double().cpu(), pe_r.grad, rtol=rt, atol=rt)
    assert torch.allclose(xbw_d.grad.double().cpu(), xb_r.grad, rtol=rt, atol=rt)
    assert torch.allclose(t_d.grad.double().cpu(), t_r.grad, rtol=1e-4, atol=1e-3 if dtype == torch.float32 else 8e-2)
    assert (t_d.grad[G - 2:] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_front_join_equals_the_per_type_route(cuda, dtype, monkeypatch):
    """ISTEncoder with ops.front_join (default) against the per-type route (embed_gelu on 'tx', torch cat + GELU on 'bd'):
    same embeddings and parameter gradients up to the rounding of one GELU implementation against the other."""
    from segger_amd import ist_encoder as IE
    from segger_amd.synthetic import SyntheticSpec
    from tests.test_gpu_model import build
    spec = SyntheticSpec(n_tx=9000, n_bd=260, k_tx=6, seed=5)
    out = {}
    for on in (True, False):
        monkeypatch.setattr(IE, "FRONT_JOIN", on)
        m, _, bcpu, _ = build(spec, cuda, dtype=dtype)
        m.eval()
        bg = bcpu.to(cuda)
        z = m(bg)
        (z["tx"].float().square().sum() * 0.5 + z["bd"].float()[:, ::2].sum()).backward()
        out[on] = (z["tx"].float().detach(), z["bd"].float().detach(),
                   {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    tol = 2e-6 if dtype == torch.float32 else 2e-2
    assert (out[True][0] - out[False][0]).abs().max().item() <= tol
    assert (out[True][1] - out[False][1]).abs().max().item() <= tol
    assert out[True][2].keys() == out[False][2].keys()
    for k, g1 in out[True][2].items():
        g0 = out[False][2][k]
        assert (g1 - g0).abs().max().item() <= (1e-4 if dtype == torch.float32 else 6e-2) * max(g0.abs().max().item(), 1e-3), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_edge_cos_argmax_at_c2_size(oracle, cuda, dtype):
    """Rows a8 / a11 at BASELINE C2 size: the C2 tile's own 2.4M tx-neighbors-bd candidate edges (1M transcripts, 10k nuclei;
    transcripts without a candidate included), random unit embeddings: cosine scores, per-transcript maximum, arg-max edge
    (first maximum = torch_scatter's CPU semantics, evaluated on the HIP scores) and assignments (with a similarity
    threshold) against the float64 oracle on the SAME storage-rounded inputs."""
    from segger_amd import ops, TX_NB_BD
    from segger_amd.graph import csr_from_coo
    from segger_amd.synthetic import SyntheticSpec, make_graph
    n_tx, n_bd, C = 1_000_000, 10_000, 64
    ei = make_graph(SyntheticSpec(n_tx=n_tx, n_bd=n_bd, k_tx=15, seed=0))[TX_NB_BD].edge_index
    src, dst = ei[0], ei[1]
    g = torch.Generator().manual_seed(8)
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, C, generator=g), dim=-1).to(dtype)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, C, generator=g), dim=-1).to(dtype)
    bd_index = torch.randperm(n_bd, generator=g).to(torch.int32) + 7
    by_src = csr_from_coo(src.to(cuda), dst.to(cuda), n_tx, n_bd)
    has = torch.bincount(src, minlength=n_tx) > 0
    assert int(ei.shape[1]) > 2_000_000 and int((~has).sum()) > 0
    for min_sim in (None, 0.05):
        seg_ref, max_ref = oracle.predict_assign(z_tx.double(), z_bd.double(), ei, bd_index, min_sim)
        sim_ref = oracle.edge_scores(z_tx.double(), z_bd.double(), ei)
        max_sim, max_eid, seg, sim = ops.edge_cos_argmax(by_src, z_tx.to(cuda), z_bd.to(cuda), dst_index=bd_index.to(cuda),
                                                         min_similarity=min_sim, return_sim=True)
        tol = 2e-6 if dtype == torch.float32 else 1e-5
        assert torch.allclose(sim.cpu().double(), sim_ref, atol=tol)
        assert torch.allclose(max_sim.cpu().double(), max_ref, atol=tol)
        _, arg_ref = oracle.scatter_max(sim.cpu(), src, n_tx)
        assert torch.equal(max_eid.cpu(), arg_ref)
        assert (max_eid.cpu()[~has] == ei.shape[1]).all() and (seg.cpu()[~has] == -1).all() and (max_sim.cpu()[~has] == 0).all()
        # assignments may differ from float64 only where the two best candidates (or the threshold) are within tol
        differ = seg.cpu() != seg_ref
        assert float(differ.float().mean()) < 1e-4, float(differ.float().mean())
