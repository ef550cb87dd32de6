"""Parity of the fused GATv2 HIP kernels (through the C ABI) with the CPU oracle.

Tolerances (stated per SURVEY.md 8(d)): the oracle runs in float64 on the SAME
(storage-rounded) inputs.
  float32 storage : |d| <= 2e-5 + 2e-5*|ref|   (v_exp_f32 / fp32 accumulation order)
  bf16 storage    : |d| <= 2e-2 + 2e-2*|ref|   (outputs are rounded to bf16: 2^-9 relative)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2e-2, 2e-2), torch.float16: (3e-3, 3e-3)}


def close(got, ref, dtype, scale=1.0, what=""):
    atol, rtol = TOL[dtype]
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    err = (got - ref).abs()
    bound = scale * (atol + rtol * ref.abs())
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.3e} (ref max {ref.abs().max():.3e})"


def random_graph(n_src, n_dst, n_edges, seed, isolated=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (n_edges,), generator=g)
    lo = 1 if (isolated and n_dst > 1) else 0          # dst 0 stays without in-edges
    dst = torch.randint(lo, n_dst, (n_edges,), generator=g)
    return torch.stack([src, dst])


def oracle_conv(oracle, xl, xr, ei, att, bias, H, gelu, keep=None, p=0.0):
    """float64 oracle on already-projected features (identity lin_l / lin_r)."""
    hc = xl.shape[1]
    eye, zero = torch.eye(hc, dtype=torch.float64), torch.zeros(hc, dtype=torch.float64)
    out, alpha = oracle.gatv2_conv(xl, xr, ei, eye, zero, eye, zero, att, bias, H,
                                   dropout_p=p, dropout_keep=keep, return_alpha=True)
    return (torch.nn.functional.gelu(out) if gelu else out), out, alpha


CASES = [
    # (H, C, n_src, n_dst, E)   low degree -> group-per-row ; high degree -> wave-per-row
    (2, 64, 300, 257, 3000),
    (2, 64, 50, 7, 1500),
    (3, 32, 120, 90, 700),
    (1, 32, 64, 33, 400),
    (4, 64, 40, 21, 300),
    (1, 64, 40, 5, 900),
    (4, 32, 64, 3, 600),
    (3, 64, 33, 40, 200),
    (2, 32, 80, 64, 0),
    # no specialised geometry -> generic one-wave-per-row kernels (any H, C <= 512, unaligned rows)
    (5, 16, 70, 40, 500),
    (1, 40, 50, 9, 700),
    (2, 128, 60, 30, 400),
    (6, 24, 40, 25, 300),
    (1, 5, 30, 12, 100),
    (3, 200, 20, 8, 90),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,C,n_src,n_dst,E", CASES)
@pytest.mark.parametrize("gelu", [False, True])
def test_gatv2_forward_backward(oracle, cuda, dtype, H, C, n_src, n_dst, E, gelu):
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    hc = H * C
    g = torch.Generator().manual_seed(H * 1000 + C + E)
    ei = random_graph(n_src, n_dst, E, seed=E + H)
    xl = torch.randn(n_src, hc, generator=g).to(dtype)
    xr = torch.randn(n_dst, hc, generator=g).to(dtype)
    att = torch.randn(hc, generator=g) * 0.3
    bias = torch.randn(hc, generator=g) * 0.1
    gy = torch.randn(n_dst, hc, generator=g).to(dtype)

    # ---- oracle (float64, autograd) -------------------------------------------
    o = [t.double().requires_grad_(True) for t in (xl, xr, att, bias)]
    y_ref, pre_ref, alpha_ref = oracle_conv(oracle, o[0], o[1], ei, o[2], o[3], H, gelu)
    y_ref.backward(gy.double())

    # ---- HIP ---------------------------------------------------------------------
    d = [t.to(cuda).requires_grad_(True) for t in (xl, xr, att, bias)]
    graph = build_edge_graph(ei.to(cuda), n_src, n_dst)
    y, alpha = ops.gatv2_aggregate(d[0], d[1], d[2], d[3], graph, H, C, apply_gelu=gelu, return_alpha=True)
    y.backward(gy.to(cuda))
    torch.cuda.synchronize()

    close(y, y_ref, dtype, what="out")
    close(alpha, alpha_ref, torch.float32 if dtype == torch.float32 else dtype, what="alpha")
    deg = max(1.0, E / max(n_dst, 1))
    close(d[0].grad, o[0].grad, dtype, scale=4.0, what="grad_xl")
    close(d[1].grad, o[1].grad, dtype, scale=4.0 * deg ** 0.5, what="grad_xr")
    # parameter gradients sum over all edges / rows: scale the absolute tolerance with sqrt(count)
    close(d[2].grad, o[2].grad, dtype, scale=4.0 * max(1.0, E) ** 0.5, what="grad_att")
    close(d[3].grad, o[3].grad, dtype, scale=4.0 * max(1.0, n_dst) ** 0.5, what="grad_bias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,n_src,n_dst,E", [(2, 64, 200, 150, 2500), (2, 64, 60, 6, 1200), (3, 32, 90, 70, 900)])
def test_gatv2_dropout_matches_explicit_mask(oracle, cuda, dtype, H, C, n_src, n_dst, E):
    """Training-mode attention dropout: the kernel's counter-based mask equals the oracle's
    restatement, forward and both backward passes regenerate the same mask."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    hc, p, seed = H * C, 0.2, 0x1234_5678_9ABC
    g = torch.Generator().manual_seed(7 + E)
    ei = random_graph(n_src, n_dst, E, seed=3)
    xl = torch.randn(n_src, hc, generator=g).to(dtype)
    xr = torch.randn(n_dst, hc, generator=g).to(dtype)
    att = torch.randn(hc, generator=g) * 0.3
    bias = torch.randn(hc, generator=g) * 0.1
    gy = torch.randn(n_dst, hc, generator=g).to(dtype)
    keep = oracle.dropout_keep_mask(seed, E, H, p)
    assert 0.7 < keep.float().mean() < 0.9

    o = [t.double().requires_grad_(True) for t in (xl, xr, att, bias)]
    y_ref, _, alpha_ref = oracle_conv(oracle, o[0], o[1], ei, o[2], o[3], H, True, keep=keep, p=p)
    y_ref.backward(gy.double())

    d = [t.to(cuda).requires_grad_(True) for t in (xl, xr, att, bias)]
    graph = build_edge_graph(ei.to(cuda), n_src, n_dst)
    y, alpha = ops.gatv2_aggregate(d[0], d[1], d[2], d[3], graph, H, C, apply_gelu=True,
                                   dropout_p=p, seed=seed, return_alpha=True)
    y.backward(gy.to(cuda))
    torch.cuda.synchronize()
    assert torch.equal((alpha.cpu() != 0), keep & (alpha_ref != 0)), "dropout mask differs"
    close(y, y_ref, dtype, what="out")
    close(alpha, alpha_ref, dtype, what="alpha")
    close(d[0].grad, o[0].grad, dtype, scale=4.0, what="grad_xl")
    close(d[1].grad, o[1].grad, dtype, scale=16.0, what="grad_xr")
    close(d[2].grad, o[2].grad, dtype, scale=4.0 * E ** 0.5, what="grad_att")


def test_gatv2_known_answer(cuda):
    """Hand-derived: 1 destination, 2 in-edges, H=1, C=32 with only channel 0/1 non-zero.
    x_r = 0, att = e_0: logits are leaky_relu(x_l[i,0]); softmax over {1.0, -1.0 -> -0.2}."""
    import math
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    H, C = 1, 32
    xl = torch.zeros(2, C); xl[0, 0], xl[0, 1] = 1.0, 2.0; xl[1, 0], xl[1, 1] = -1.0, 4.0
    xr = torch.zeros(1, C)
    att = torch.zeros(C); att[0] = 1.0
    bias = torch.zeros(C); bias[1] = 0.5
    ei = torch.tensor([[0, 1, 1], [0, 0, 0]])            # duplicate edge 1->0 counts twice
    e = [1.0, -0.2, -0.2]
    w = [math.exp(v) for v in e]
    a = [v / sum(w) for v in w]
    want0 = a[0] * 1.0 + (a[1] + a[2]) * -1.0
    want1 = a[0] * 2.0 + (a[1] + a[2]) * 4.0 + 0.5
    graph = build_edge_graph(ei.to(cuda), 2, 1)
    out, alpha = ops.gatv2_aggregate(xl.to(cuda), xr.to(cuda), att.to(cuda), bias.to(cuda), graph, H, C, return_alpha=True)
    assert abs(out[0, 0].item() - want0) < 1e-6 and abs(out[0, 1].item() - want1) < 1e-6
    assert torch.allclose(alpha.cpu().flatten(), torch.tensor(a, dtype=torch.float32), atol=1e-6)


def test_gatv2_edge_order_invariance_and_isolated(cuda):
    """Permuting the COO edge list changes nothing (up to fp32 summation order); a destination
    without in-edges returns the bias."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    H, C, n_src, n_dst, E = 2, 64, 100, 60, 900
    g = torch.Generator().manual_seed(0)
    ei = random_graph(n_src, n_dst, E, seed=11)
    perm = torch.randperm(E, generator=g)
    xl, xr = torch.randn(n_src, H * C, generator=g), torch.randn(n_dst, H * C, generator=g)
    att, bias = torch.randn(H * C, generator=g) * 0.3, torch.randn(H * C, generator=g)
    outs = []
    for e in (ei, ei[:, perm]):
        graph = build_edge_graph(e.to(cuda), n_src, n_dst)
        outs.append(ops.gatv2_aggregate(xl.to(cuda), xr.to(cuda), att.to(cuda), bias.to(cuda), graph, H, C))
    assert torch.allclose(outs[0], outs[1], atol=1e-5, rtol=1e-5)
    assert torch.equal(outs[0][0].cpu(), bias)          # dst 0 is isolated by construction


@pytest.mark.parametrize("window", [4, 16, 64])
def test_row_order_is_a_permutation_and_changes_nothing(cuda, window):
    """segger_csr_row_order: a permutation of the rows, sorted by descending degree inside every window; forward and
    backward results with the order attached are bit-identical to the natural order (rows are independent)."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    H, C, n_src, n_dst, E = 2, 64, 500, 333, 5000
    g = torch.Generator().manual_seed(3)
    ei = random_graph(n_src, n_dst, E, seed=5).to(cuda)
    graph = build_edge_graph(ei, n_src, n_dst)
    assert graph.by_dst.order is None                                      # off by default (graph.ROW_ORDER_WINDOW)
    xl = torch.randn(n_src, H * C, generator=g).to(cuda).requires_grad_(True)
    xr = torch.randn(n_dst, H * C, generator=g).to(cuda).requires_grad_(True)
    att = (torch.randn(H * C, generator=g) * 0.3).to(cuda).requires_grad_(True)
    w = torch.randn(n_dst, H * C, generator=g).to(cuda)

    def run():
        out = ops.gatv2_aggregate(xl, xr, att, None, graph, H, C, apply_gelu=True, dropout_p=0.2, seed=9)
        grads = torch.autograd.grad((out * w).sum(), (xl, xr, att))
        return (out.detach(),) + grads
    base = run()
    graph.by_dst.balanced_order(window)
    graph.by_src.balanced_order(window)
    order = graph.by_dst.order.long().cpu()
    assert torch.equal(order.sort().values, torch.arange(n_dst))
    deg = (graph.by_dst.indptr[1:] - graph.by_dst.indptr[:-1]).cpu()[order]
    for b in range(0, n_dst, window):
        d = deg[b:b + window]
        assert bool((d[:-1] >= d[1:]).all()) and int(order[b:b + window].min()) >= b and int(order[b:b + window].max()) < b + window
    again = run()
    for a, b_ in zip(base[:3], again[:3]):                 # per-row results: bit-identical
        assert torch.equal(a, b_)
    # grad_att: block partials regroup an fp32 sum of ~5000 terms of either sign
    assert (base[3] - again[3]).abs().max() <= 1e-5 * base[3].abs().max() + 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,n_src,n_dst,E", [(2, 64, 4000, 60, 2500), (2, 64, 900, 400, 700), (3, 32, 500, 7, 480),
                                               (5, 16, 300, 20, 250)])
def test_unique_sources_need_no_by_source_view(oracle, cuda, dtype, H, C, n_src, n_dst, E):
    """tx-belongs-bd: every source has at most one out-edge, so the destination pass stores grad_x_l itself
    (segger_gatv2_bwd_args.src_unique; by_src is never sorted).  Gradients equal the two-pass path's to the last bit or two and
    the oracle's within the usual tolerance; a graph with a repeated source silently falls back to the sort."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    g = torch.Generator().manual_seed(E)
    src = torch.randperm(n_src, generator=g)[:E]                          # distinct sources
    dst = torch.randint(0, n_dst, (E,), generator=g)
    ei = torch.stack([src, dst])
    xl = torch.randn(n_src, H * C, generator=g).to(dtype)
    xr = torch.randn(n_dst, H * C, generator=g).to(dtype)
    att, bias = torch.randn(H * C, generator=g) * 0.3, torch.randn(H * C, generator=g) * 0.1
    w = torch.randn(n_dst, H * C, generator=g)

    def run(graph, p):
        leaves = [t.to(cuda).requires_grad_(True) for t in (xl, xr, att, bias)]
        out = ops.gatv2_aggregate(*leaves, graph, H, C, apply_gelu=True, dropout_p=p, seed=3)
        (out.float() * w.to(cuda)).sum().backward()
        return [out.detach()] + [t.grad for t in leaves]
    lazy = build_edge_graph(ei.to(cuda), n_src, n_dst, need_by_src="lazy")
    full = build_edge_graph(ei.to(cuda), n_src, n_dst)
    assert lazy.by_src is None
    def same(a, b):
        # fp32: the one-pass form rounds a*g + rest once where the two-pass form rounds a*g first (last-bit differences).
        # bf16: the two-pass form reads grad_pre back ROUNDED to bf16 (2^-9 relative) where the one-pass form still holds
        # it in fp32, so the results differ by a few bf16 ulps -- in favour of the one-pass form
        rt = 2e-6 if dtype == torch.float32 else 2.0 ** -5
        for x, y in zip(a, b):
            assert torch.allclose(x.float(), y.float(), rtol=rt, atol=rt * float(y.float().abs().max()) * 0.25 + 1e-9)
    for p in (0.0, 0.3):
        same(run(lazy, p), run(full, p))
    assert lazy.src_unique() and (lazy.by_src is None) == (H != 5)          # (5, 16) has no specialised kernel: sorted lazily
    # against the oracle (no dropout)
    got = run(lazy, 0.0)
    o = [t.double().requires_grad_(True) for t in (xl, xr, att, bias)]
    ref, _, _ = oracle_conv(oracle, o[0], o[1], ei, o[2], o[3], H, True)
    (ref * w.double()).sum().backward()
    close(got[0], ref, dtype, what="out")
    close(got[1], o[0].grad, dtype, scale=4.0, what="grad_xl")
    close(got[2], o[1].grad, dtype, scale=4.0 * max(1.0, E / n_dst) ** 0.5, what="grad_xr")
    # a repeated source: the flag says so and the by-source view is built on first use
    ei2 = ei.clone(); ei2[0, 1] = ei2[0, 0]
    dup = build_edge_graph(ei2.to(cuda), n_src, n_dst, need_by_src="lazy")
    full2 = build_edge_graph(ei2.to(cuda), n_src, n_dst)
    same(run(dup, 0.0), run(full2, 0.0))                                   # same kernels, same views
    assert not dup.src_unique() and dup.by_src is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,C,n_tx,n_bd,k,frac", [(2, 64, 3000, 40, 6, 0.9), (2, 64, 900, 300, 5, 0.5), (4, 32, 700, 9, 4, 1.0),
                                                  (1, 32, 500, 6, 3, 0.8), (5, 16, 300, 5, 4, 0.9)])
@pytest.mark.parametrize("p", [0.0, 0.3, -0.3])          # -0.3: dropout 0.3 by the counter hash (no precomputed bit planes)
def test_hetero_layer_pair_launches_equal_single_edge_types(cuda, dtype, H, C, n_tx, n_bd, k, frac, p):
    """One hetero layer (ist_encoder.py:109-134: tx-neighbors-tx + tx-belongs-bd) through the merged launches
    (segger_gatv2_fwd_pair; segger_gatv2_bwd_pair: zero fill in the tx-neighbors-tx destination pass, its source pass in
    one launch with the tx-belongs-bd pass) against the two edge types run one by one: the same kernels' bodies, so the
    per-row results agree to the last bit or two; (900, 300) has too few transcripts per boundary for the wave-per-row form and
    (5, 16) no specialised kernel -- both take the library's two-call route."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    g = torch.Generator().manual_seed(n_tx + k)
    hc = H * C
    dst = torch.arange(n_tx).repeat_interleave(k)
    src = torch.randint(0, n_tx, (n_tx * k,), generator=g)
    ei_tt = torch.stack([src, dst])
    members = torch.randperm(n_tx, generator=g)[: int(frac * n_tx)]             # each transcript in at most one boundary
    ei_tb = torch.stack([members, torch.randint(0, n_bd, (members.numel(),), generator=g)])
    g_tt = build_edge_graph(ei_tt.to(cuda), n_tx, n_tx)
    g_tb = build_edge_graph(ei_tb.to(cuda), n_tx, n_bd, need_by_src="lazy")
    xp_tx = torch.randn(n_tx, 3 * hc, generator=g).to(dtype)
    xp_bd = torch.randn(n_bd, hc, generator=g).to(dtype)
    vec = lambda s_: torch.randn(hc, generator=g) * s_
    att_tt, bias_tt, att_tb, bias_tb = vec(0.3), vec(0.1), vec(0.3), vec(0.1)
    w_tx, w_bd = torch.randn(n_tx, hc, generator=g).to(cuda), torch.randn(n_bd, hc, generator=g).to(cuda)
    bits_tt = bits_tb = None
    use_bits, p = p > 0, abs(p)
    if use_bits:
        bits_tt = (ops.dropout_bits(g_tt.by_dst, H, p, [11])[0], ops.dropout_bits(g_tt.by_src, H, p, [11])[0])
        bits_tb = (ops.dropout_bits(g_tb.by_dst, H, p, [12])[0], None)

    def leaves():
        return [t.to(cuda).requires_grad_(True) for t in (xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb)]

    def merged():
        L = leaves()
        y_tx, y_bd, _ = ops.hetero_gat_layer(*L, g_tt, g_tb, H, C, dropout_p=p, seed_tt=11, seed_tb=12,
                                             bits_tt=bits_tt, bits_tb=bits_tb)
        ((y_tx.float() * w_tx).sum() + (y_bd.float() * w_bd).sum()).backward()
        return [y_tx.detach(), y_bd.detach()] + [t.grad for t in L]

    def one_by_one():
        L = leaves()
        xp, xb = L[0], L[1]
        y_tx = ops.gatv2_aggregate(xp[:, :hc], xp[:, hc:2 * hc], L[2], L[3], g_tt, H, C, apply_gelu=True, dropout_p=p,
                                   seed=11, keep_bits=bits_tt)
        y_bd = ops.gatv2_aggregate(xp[:, 2 * hc:], xb, L[4], L[5], g_tb, H, C, apply_gelu=True, dropout_p=p, seed=12,
                                   keep_bits=bits_tb)
        ((y_tx.float() * w_tx).sum() + (y_bd.float() * w_bd).sum()).backward()
        return [y_tx.detach(), y_bd.detach()] + [t.grad for t in L]

    def rows_same(x, y, i):
        # the merged backward builds the tx-belongs-bd pass at 2 rows in flight instead of 4: the compiler contracts /
        # orders a few fp32 operations differently, so allow the last bit (fp32) or one rounding flip (16-bit storage)
        rt = 2e-6 if dtype == torch.float32 else 2.0 ** -7
        assert torch.allclose(x.float(), y.float(), rtol=rt, atol=rt * float(y.float().abs().max()) * 0.25 + 1e-9), \
            f"tensor {i} differs: {(x.float() - y.float()).abs().max()}"

    a, b = merged(), one_by_one()
    for i, (x, y) in enumerate(zip(a[:2], b[:2])):            # outputs: the same kernel bodies
        assert torch.equal(x, y), f"output {i} differs: {(x.float() - y.float()).abs().max()}"
    for i, (x, y) in enumerate(zip(a[2:4], b[2:4])):          # the row gradients
        rows_same(x, y, i)
    for i, (x, y) in enumerate(zip(a[4:], b[4:])):            # att / bias gradients: block partials may regroup an fp32 sum
        assert (x - y).abs().max() <= 1e-5 * y.abs().max() + 1e-5, f"parameter gradient {i}"
    if ops._BWD_PAIR:                                         # and the switch really selects the two-call route
        ops._BWD_PAIR = 0
        try:
            c = merged()
        finally:
            ops._BWD_PAIR = 1
        for i, (x, y) in enumerate(zip(a[:4], c[:4])):
            rows_same(x, y, i)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,n_src,n_dst,E", [(2, 64, 300, 257, 3000), (2, 64, 50, 7, 1500), (4, 32, 64, 40, 900), (3, 64, 90, 50, 400)])
def test_dropout_bit_planes_equal_the_hash(cuda, dtype, H, C, n_src, n_dst, E):
    """segger_dropout_bits precomputes keep(e, h) of several layers as bit planes per CSR view; forward, both backward
    passes (and the one-pass backward) fed the planes give bit-identical results to hashing inside the kernels."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    g = torch.Generator().manual_seed(E + H)
    ei = random_graph(n_src, n_dst, E, seed=E).to(cuda)
    graph = build_edge_graph(ei, n_src, n_dst)
    xl = torch.randn(n_src, H * C, generator=g).to(dtype).to(cuda)
    xr = torch.randn(n_dst, H * C, generator=g).to(dtype).to(cuda)
    att, bias = (torch.randn(H * C, generator=g) * 0.3).to(cuda), (torch.randn(H * C, generator=g) * 0.1).to(cuda)
    w = torch.randn(n_dst, H * C, generator=g).to(cuda)
    step = torch.tensor([512], dtype=torch.int64, device=cuda)
    seeds = [0, 2, 4, 6]
    planes_d = ops.dropout_bits(graph.by_dst, H, 0.25, seeds, step)
    planes_s = ops.dropout_bits(graph.by_src, H, 0.25, seeds, step)
    assert planes_d.shape == (4, E) and planes_d.dtype == torch.uint8
    # the planes are the oracle-tested mask itself: plane l, slot s, bit h == keep(eid[s], h; seed_l + step)
    keep = ops.dropout_bits(graph.by_dst, H, 0.25, [seeds[2]], step)[0]
    assert torch.equal(keep, planes_d[2]) and 0.6 < float(((planes_d[1] >> 1) & 1).float().mean()) < 0.9
    # same edge, both views: planes agree through the edge ids
    full_d = torch.empty(E, dtype=torch.uint8, device=cuda); full_d[graph.by_dst.eid.long()] = planes_d[3]
    full_s = torch.empty(E, dtype=torch.uint8, device=cuda); full_s[graph.by_src.eid.long()] = planes_s[3]
    assert torch.equal(full_d, full_s)
    # all views of a step in one launch (segger_dropout_bits_many) == one launch per view; the step counter advanced and
    # snapshotted by one launch (segger_step_advance)
    many = ops.dropout_bits_many([(graph.by_dst, seeds), (graph.by_src, seeds), (graph.by_dst, [seeds[2]])], H, 0.25, step)
    assert torch.equal(many[0], planes_d) and torch.equal(many[1], planes_s) and torch.equal(many[2][0], keep)
    ctr = step.clone()
    snap = ops.step_advance(ctr, 256)
    assert int(ctr) == 768 and int(snap) == 768 and snap.data_ptr() != ctr.data_ptr()

    def run(bits, li):
        leaves = [t.clone().requires_grad_(True) for t in (xl, xr, att, bias)]
        out = ops.gatv2_aggregate(*leaves, graph, H, C, apply_gelu=True, dropout_p=0.25, seed=(seeds[li], step), keep_bits=bits)
        (out.float() * w).sum().backward()
        return [out.detach()] + [t.grad for t in leaves]
    for li in (0, 3):
        a, b = run(None, li), run((planes_d[li], planes_s[li]), li)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        c = run((planes_d[li], None), li)                   # planes for one view only: the other pass hashes
        for x, y in zip(a, c):
            assert torch.equal(x, y)


def test_rejects_bad_arguments(cuda):
    from segger_amd import ops, _lib
    from segger_amd.graph import build_edge_graph
    ei = torch.tensor([[0, 1], [0, 5]])
    with pytest.raises(IndexError):
        build_edge_graph(ei.to(cuda), 2, 2)
    graph = build_edge_graph(torch.tensor([[0, 1], [0, 1]]).to(cuda), 2, 2)
    x = torch.randn(2, 600, device=cuda)
    with pytest.raises(_lib.SeggerAmdError, match="exceeds the generic kernel's limit"):
        ops.gatv2_aggregate(x, x, torch.randn(600, device=cuda), None, graph, 1, 600)
    x = torch.randn(2, 40, device=cuda)
    with pytest.raises(_lib.SeggerAmdError, match="no CPU fallback"):
        ops.gatv2_aggregate(x.cpu(), x.cpu(), torch.randn(40), None, graph, 1, 40)


def test_dropout_mask_known_answer_on_device(cuda):
    """The kernels' dropout stream against the literal bits of tests/test_oracle.py (seed 5, H = 2, p = 0.2): the
    attention output is exactly zero where, and only where, the pinned mask drops the edge."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    want = "11111011111111110110110001010111111010111110001100010011001111011101111111111111"
    keep = torch.tensor([int(c) for c in want], dtype=torch.bool).reshape(40, 2)
    H, C, n = 2, 64, 12
    g = torch.Generator().manual_seed(0)
    ei = torch.stack([torch.randint(0, n, (40,), generator=g), torch.randint(0, n, (40,), generator=g)])
    graph = build_edge_graph(ei.to(cuda), n, n)
    x = torch.randn(n, H * C, generator=g).to(cuda)
    att = (torch.randn(H * C, generator=g) * 0.3).to(cuda)
    _, alpha = ops.gatv2_aggregate(x, x, att, None, graph, H, C, dropout_p=0.2, seed=5, return_alpha=True)
    assert torch.equal((alpha != 0).cpu(), keep)


# ---- the roofline kernels at BASELINE's full C2 size against the oracle (round-5 review, item 3) -----------------------
_C2 = {}


def _c2_tile(cuda):
    """The bench's own C2 tile (1M transcripts, k = 15 kNN: variable in-degrees, so the window-64 visiting order of the
    destination pass is active), built once per session."""
    if "ei" not in _C2:
        from segger_amd import TX_TX
        from segger_amd.graph import build_edge_graph
        from segger_amd.synthetic import SyntheticSpec, make_graph
        n = 1_000_000
        b = make_graph(SyntheticSpec(n_tx=n, n_bd=n // 100, k_tx=15, seed=0))
        ei = b[TX_TX].edge_index.to(cuda)
        _C2.update(n=n, ei=ei, graph=build_edge_graph(ei, n, n))
    return _C2["n"], _C2["ei"], _C2["graph"]


def _subgraph(ei, rows_flag_dst):
    """Edges whose destination is flagged -> (edge ids, unique sources, local source index, unique destinations, local
    destination index); everything on the device."""
    sel = rows_flag_dst[ei[1]].nonzero().squeeze(1)
    src, dst = ei[0, sel], ei[1, sel]
    us, ls = torch.unique(src, return_inverse=True)
    ud, ld = torch.unique(dst, return_inverse=True)
    return sel, us, ls, ud, ld


@pytest.mark.parametrize("dropout", [0.0, 0.2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_c2_layer_against_oracle_on_sampled_rows(oracle, cuda, dtype, dropout):
    """One tx-neighbors-tx layer at C2 size (1M rows, 15M edges, general att and bias, GELU, dropout off and on with the
    step's bit planes): 2000 sampled destination rows (first, last and the highest in-degree row among them) and 1000
    sampled source rows are checked against ``oracle.gatv2_conv`` in float64 on the sub-graphs that determine them --
    `out`, `alpha` and `grad_xr` of a destination depend on its in-edges only; `grad_xl` of a source on ALL in-edges of
    every destination it points to.  Tolerances: the per-kernel ones of this file."""
    from segger_amd import ops
    n, ei, graph = _c2_tile(cuda)
    H, C = 2, 64
    hc, E = H * C, int(ei.shape[1])
    assert graph.by_dst.order is not None and graph.by_src.order is None        # graph.ROW_ORDER_WINDOW_DST at this size
    gen = torch.Generator(device=cuda).manual_seed(11)
    xl = torch.randn(n, hc, device=cuda, generator=gen).to(dtype).requires_grad_(True)
    xr = torch.randn(n, hc, device=cuda, generator=gen).to(dtype).requires_grad_(True)
    att = (torch.randn(hc, device=cuda, generator=gen) * 0.3).requires_grad_(True)
    bias = (torch.randn(hc, device=cuda, generator=gen) * 0.1).requires_grad_(True)
    gy = torch.randn(n, hc, device=cuda, generator=gen).to(dtype)
    seed = 0x5EED_0000_0C2
    bits = None
    if dropout > 0:
        bits = (ops.dropout_bits(graph.by_dst, H, dropout, [seed])[0], ops.dropout_bits(graph.by_src, H, dropout, [seed])[0])
    y, alpha = ops.gatv2_aggregate(xl, xr, att, bias, graph, H, C, apply_gelu=True, dropout_p=dropout, seed=seed,
                                   return_alpha=True, keep_bits=bits)
    y.backward(gy)
    torch.cuda.synchronize()

    deg = torch.bincount(ei[1], minlength=n)
    pick = torch.randperm(n, device=cuda, generator=gen)
    dsts = torch.cat([pick[:1997], torch.tensor([0, n - 1], device=cuda), deg.argmax().view(1)]).unique()
    srcs = pick[2000:3000]
    keep_all = oracle.dropout_keep_mask(seed, E, H, dropout) if dropout > 0 else None

    def run_oracle(flag):
        sel, us, ls, ud, ld = _subgraph(ei, flag)
        o = [t.detach().double().cpu().requires_grad_(True) for t in (xl[us], xr[ud], att, bias)]
        sub = torch.stack([ls, ld]).cpu()
        keep = keep_all[sel.cpu()] if keep_all is not None else None
        eye, zero = torch.eye(hc, dtype=torch.float64), torch.zeros(hc, dtype=torch.float64)
        pre, a = oracle.gatv2_conv(o[0], o[1], sub, eye, zero, eye, zero, o[2], o[3], H, dropout_p=dropout,
                                   dropout_keep=keep, return_alpha=True)
        out = torch.nn.functional.gelu(pre)
        out.backward(gy[ud].double().cpu())
        return sel, us, ud, out, a, o

    # ---- destinations: out, alpha, grad_xr -------------------------------------------------------------------------------
    flag = torch.zeros(n, dtype=torch.bool, device=cuda); flag[dsts] = True
    sel, us, ud, out_ref, a_ref, o = run_oracle(flag)
    assert torch.equal(ud, dsts) and int(deg[dsts].max()) == int(deg.max()) and int(deg[dsts].min()) >= 0
    close(y[ud], out_ref, dtype, what="out (C2 rows)")
    close(alpha[sel], a_ref, torch.float32 if dtype == torch.float32 else dtype, what="alpha (C2 edges)")
    if dropout > 0:
        assert torch.equal(alpha[sel].cpu() != 0, keep_all[sel.cpu()] & (a_ref != 0)), "dropout mask differs at C2 size"
    k_eff = float(deg[dsts].float().mean())
    close(xr.grad[ud], o[1].grad, dtype, scale=4.0 * max(1.0, k_eff) ** 0.5, what="grad_xr (C2 rows)")

    # ---- sources: grad_xl needs every in-edge of every destination a sampled source points to ----------------------------
    sflag = torch.zeros(n, dtype=torch.bool, device=cuda); sflag[srcs] = True
    touched = ei[1, sflag[ei[0]].nonzero().squeeze(1)]
    flag = torch.zeros(n, dtype=torch.bool, device=cuda); flag[touched] = True
    sel, us, ud, out_ref, a_ref, o = run_oracle(flag)
    local = torch.searchsorted(us, srcs.sort().values)
    assert torch.equal(us[local], srcs.sort().values)               # every sampled source is a source of the sub-graph (self edge)
    close(xl.grad[srcs.sort().values], o[0].grad[local.cpu()], dtype, scale=4.0, what="grad_xl (C2 rows)")
    close(y[ud], out_ref, dtype, what="out (rows around the sampled sources)")


@pytest.mark.parametrize("dropout", [0.0, 0.2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_c2_belongs_layer_against_oracle(oracle, cuda, dtype, dropout):
    """The OTHER edge type of a C2 layer at full size -- tx-belongs-bd: 1M source transcripts, 10k boundary rows of 40-400
    in-edges, the wave-per-row forward and the ONE-PASS backward (every transcript lies in at most one boundary:
    ``src_unique``, grad_xl stored by the destination pass) -- against ``oracle.gatv2_conv`` in float64 on the WHOLE edge
    type (0.39M edges: the oracle handles it in seconds; sources compacted to the transcripts that have an edge, the
    others must come back with a zero gradient)."""
    from segger_amd import ops, TX_BD
    from segger_amd.graph import build_edge_graph
    from segger_amd.synthetic import SyntheticSpec, make_graph
    if "tb" not in _C2:
        n = 1_000_000
        b = make_graph(SyntheticSpec(n_tx=n, n_bd=n // 100, k_tx=15, seed=0))
        _C2["tb"] = b[TX_BD].edge_index.to(cuda)
    ei = _C2["tb"]
    n_src, n_dst, H, C = 1_000_000, 10_000, 2, 64
    hc, E = H * C, int(ei.shape[1])
    graph = build_edge_graph(ei, n_src, n_dst, need_by_src="lazy")
    assert graph.src_unique() and E > 300_000
    gen = torch.Generator(device=cuda).manual_seed(23)
    xl = torch.randn(n_src, hc, device=cuda, generator=gen).to(dtype).requires_grad_(True)
    xr = torch.randn(n_dst, hc, device=cuda, generator=gen).to(dtype).requires_grad_(True)
    att = (torch.randn(hc, device=cuda, generator=gen) * 0.3).requires_grad_(True)
    bias = (torch.randn(hc, device=cuda, generator=gen) * 0.1).requires_grad_(True)
    gy = torch.randn(n_dst, hc, device=cuda, generator=gen).to(dtype)
    seed = 0xB07D_0C2
    bits = (ops.dropout_bits(graph.by_dst, H, dropout, [seed])[0], None) if dropout > 0 else None
    y, alpha = ops.gatv2_aggregate(xl, xr, att, bias, graph, H, C, apply_gelu=True, dropout_p=dropout, seed=seed,
                                   return_alpha=True, keep_bits=bits)
    y.backward(gy)
    torch.cuda.synchronize()

    us, ls = torch.unique(ei[0], return_inverse=True)
    o = [t.detach().double().cpu().requires_grad_(True) for t in (xl[us], xr, att, bias)]
    keep = oracle.dropout_keep_mask(seed, E, H, dropout) if dropout > 0 else None
    eye, zero = torch.eye(hc, dtype=torch.float64), torch.zeros(hc, dtype=torch.float64)
    pre, a_ref = oracle.gatv2_conv(o[0], o[1], torch.stack([ls, ei[1]]).cpu(), eye, zero, eye, zero, o[2], o[3], H,
                                   dropout_p=dropout, dropout_keep=keep, return_alpha=True)
    torch.nn.functional.gelu(pre).backward(gy.double().cpu())
    close(y, torch.nn.functional.gelu(pre), dtype, what="out (tx-belongs-bd, C2)")
    close(alpha, a_ref, torch.float32 if dtype == torch.float32 else dtype, what="alpha (tx-belongs-bd, C2)")
    deg = float(E) / n_dst
    close(xl.grad[us], o[0].grad, dtype, scale=4.0, what="grad_xl (tx-belongs-bd, C2)")
    no_edge = torch.ones(n_src, dtype=torch.bool, device=cuda); no_edge[us] = False
    assert float(xl.grad[no_edge].abs().max()) == 0.0, "transcripts without a boundary must get a zero gradient"
    close(xr.grad, o[1].grad, dtype, scale=4.0 * deg ** 0.5, what="grad_xr (tx-belongs-bd, C2)")
    close(att.grad, o[2].grad, dtype, scale=4.0 * E ** 0.5, what="grad_att (tx-belongs-bd, C2)")
    close(bias.grad, o[3].grad, dtype, scale=4.0 * n_dst ** 0.5, what="grad_bias (tx-belongs-bd, C2)")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_lds_gather_forward_is_bit_identical(cuda, dtype):
    """Opt-in block tables (``EdgeCSR.block_tables()`` -> ``gatv2_fwd_lds_kernel``): every workgroup stages the DISTINCT source
    rows of its 16 destination rows in LDS once and gathers from there.  Same values, same order of accumulation: outputs,
    pre-activations and softmax statistics equal the plain kernel's bit for bit -- on the C2 tile (dropout bit planes on) and
    on a small random graph with empty rows; a graph whose workgroups gather more than 128 distinct rows gets no tables."""
    from segger_amd import ops
    from segger_amd.graph import build_edge_graph
    H, C = 2, 64
    hc = H * C
    n, ei, _ = _c2_tile(cuda)
    small = random_graph(5000, 3001, 40_000, seed=4).to(cuda)
    small = small[:, torch.argsort(small[1] // 16 * 100_000 + small[0] % 97)]          # (few distinct sources per 16 rows)
    small[0] = (small[1] // 16) * 8 % 4900 + small[0] % 97
    for n_src, n_dst, edges, p in ((n, n, ei, 0.2), (5000, 3001, small, 0.0)):
        g0 = build_edge_graph(edges, n_src, n_dst)
        g1 = build_edge_graph(edges, n_src, n_dst)
        g1.by_dst.block_tables()
        assert g1.by_dst.tables is not None and g0.by_dst.tables is None
        gen = torch.Generator(device=cuda).manual_seed(2)
        xl = torch.randn(n_src, hc, device=cuda, generator=gen).to(dtype)
        xr = torch.randn(n_dst, hc, device=cuda, generator=gen).to(dtype)
        att = torch.randn(hc, device=cuda, generator=gen) * 0.3
        bias = torch.randn(hc, device=cuda, generator=gen) * 0.1
        res = []
        for g in (g0, g1):
            out = torch.empty(n_dst, hc, dtype=dtype, device=cuda)
            pre = torch.empty_like(out)
            lse = torch.empty(n_dst, H, device=cuda)
            bits = ops.dropout_bits(g.by_dst, H, p, [9])[0] if p > 0 else None
            ops.gatv2_fwd_launch(g.by_dst, xl, xr, att, bias, H, C, out, pre=pre, lse=lse, apply_gelu=True, dropout_p=p, seed=9,
                                 keep_bits=bits)
            res.append((out, pre, lse))
        torch.cuda.synchronize()
        for a, b in zip(*res):
            assert torch.equal(a, b)
    wide = torch.stack([torch.randperm(4000, device=cuda)[:3200], torch.arange(3200, device=cuda) % 16])   # 200 sources per row
    gw = build_edge_graph(wide, 4000, 16)
    gw.by_dst.block_tables()
    assert gw.by_dst.tables is None
