"""N2 on the device: the tile batcher (segger_amd/tiles.py) run on ``cuda`` tensors against naive restatements
computed on the HOST copy of the same graph -- partition vs brute force (reference data/partition/dataset.py:340-579
as used by data/tile_dataset.py:13-153), ``PredictTiles`` vs a bounding-box filter and ``PredictTileIndex`` vs
``PredictTiles`` (data/tile_dataset.py:156-264), the bin-packed sampler over a device partition vs the one over the
host partition (packers pinned by the reference's own vectors in tests/test_tiles.py), the slide-level CSR slices of
``TilePartition.build_csr`` vs a stable argsort on the host, and a bucket-padded ``segger_stage``d batch of the
captured training step vs ``TilePartition.batch``.  Indices are compared bit-exactly on the host."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from segger_amd.hetero import TX_BD, TX_NB_BD, TX_TX, collate          # noqa: E402
from segger_amd.synthetic import SyntheticSpec, make_graph             # noqa: E402
from segger_amd import tiles as T                                      # noqa: E402

ETS = (TX_TX, TX_BD, TX_NB_BD)


@pytest.fixture(scope="module")
def graphs(cuda):
    g = make_graph(SyntheticSpec(n_tx=3000, n_bd=90, k_tx=6, seed=4))
    for nt in ("tx", "bd"):
        del g[nt]["mask"]                       # the fit mask is what the tile partition adds
    return g, g.to(cuda)


def all_pos(g):
    return torch.cat([g["tx"].pos, g["bd"].pos])


def naive_csr(rows, cols, n_rows):
    """Host restatement of the by-row view: stable order, eid = original COO position."""
    order = torch.argsort(rows, stable=True)
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(rows, minlength=n_rows).cumsum(0)])
    return indptr, cols[order].to(torch.int32), order.to(torch.int32)


def test_partition_on_device_matches_brute_force(graphs):
    host, dev = graphs
    tiling = T.SquareTiling(all_pos(dev), 40.0)
    part = T.partition_by_tiling(dev, tiling, margin=3.0)
    assert part.data["tx"]["pos"].is_cuda and part.data[TX_TX].edge_index.is_cuda
    h_tiling = T.SquareTiling(all_pos(host), 40.0)
    labels = {nt: h_tiling.label(host[nt].pos) for nt in ("tx", "bd")}
    n_t = len(tiling)
    assert n_t == len(h_tiling) and sum(part.node_sizes["tx"].tolist()) == 3000
    for t in range(n_t):
        tile = part.tile(t)
        for nt in ("tx", "bd"):
            ids = (labels[nt] == t).nonzero().squeeze(1)                     # stable order inside a tile
            assert torch.equal(tile[nt].index.long().cpu(), host[nt].index[ids].long())
            assert torch.equal(tile[nt].pos.cpu(), host[nt].pos[ids])
            assert torch.equal(tile[nt]["mask"].cpu(), h_tiling.mask(host[nt].pos[ids], 3.0))
            assert bool((tile[nt]["batch"] == 0).all())
        for et in ETS:
            s, _, d = et
            ei = host[et].edge_index
            keep = (labels[s][ei[0]] == t) & (labels[d][ei[1]] == t)         # intra-tile edges only, original order
            want = torch.stack([host[s].index[ei[0, keep]].long(), host[d].index[ei[1, keep]].long()])
            got_ei = tile[et].edge_index.cpu()
            got = torch.stack([tile[s].index.long().cpu()[got_ei[0]], tile[d].index.long().cpu()[got_ei[1]]])
            assert torch.equal(got, want), (t, et)
    # a batch assembled on the device partition == collation of the individual tiles of the HOST partition
    h_part = T.partition_by_tiling(host, h_tiling, margin=3.0)
    ids = [3, 0, n_t - 1]
    b1, b2 = part.batch(ids), collate([h_part.tile(i) for i in ids])
    assert b1.num_graphs == 3
    for nt in ("tx", "bd"):
        for a in ("x", "pos", "index", "mask", "cluster", "batch"):
            assert torch.equal(b1[nt][a].cpu(), b2[nt][a]), (nt, a)
    for et in ETS:
        assert torch.equal(b1[et].edge_index.cpu(), b2[et].edge_index)


def test_predict_tiles_on_device_match_bbox_filter(graphs):
    host, dev = graphs
    tiling = T.SquareTiling(all_pos(dev), 30.0)
    ds = T.PredictTiles(dev, tiling.tiles, margin=4.0)
    seen = torch.zeros(3000, dtype=torch.long)
    pos, bpos = host["tx"].pos, host["bd"].pos
    for i in range(len(ds)):
        x0, y0, x1, y1 = ds.tiles[i].tolist()
        t = ds[i]
        assert t["tx"].pos.is_cuda
        outer = (pos[:, 0] >= x0 - 4) & (pos[:, 0] < x1 + 4) & (pos[:, 1] >= y0 - 4) & (pos[:, 1] < y1 + 4)
        assert torch.equal(t["tx"].index.cpu(), host["tx"].index[outer])
        tp = t["tx"].pos.cpu()
        inner = (tp[:, 0] >= x0) & (tp[:, 0] <= x1) & (tp[:, 1] >= y0) & (tp[:, 1] <= y1)
        assert torch.equal(t["tx"].predict_mask.cpu(), inner)
        seen[t["tx"].index.cpu()[inner]] += 1
        bd_outer = (bpos[:, 0] >= x0 - 4) & (bpos[:, 0] < x1 + 4) & (bpos[:, 1] >= y0 - 4) & (bpos[:, 1] < y1 + 4)
        for et in ETS:
            s, _, d = et
            g_ei = host[et].edge_index
            o_s, o_d = (outer if s == "tx" else bd_outer), (outer if d == "tx" else bd_outer)
            keep = o_s[g_ei[0]] & o_d[g_ei[1]]
            want = torch.stack([host[s].index[g_ei[0, keep]].long(), host[d].index[g_ei[1, keep]].long()])
            ei = t[et].edge_index.cpu()
            got = torch.stack([t[s].index.long().cpu()[ei[0]], t[d].index.long().cpu()[ei[1]]])
            assert torch.equal(got, want), (i, et)                           # same edges, same (original) order
    assert bool((seen >= 1).all())


@pytest.mark.parametrize("side,margin", [(30.0, 4.0), (17.0, 17.0), (20.0, 10.0), (12.0, 0.0)])
def test_predict_tile_index_on_device_equals_host_predict_tiles(graphs, side, margin):
    host, dev = graphs
    tiling = T.SquareTiling(all_pos(dev), side)
    slow = T.PredictTiles(host, T.SquareTiling(all_pos(host), side).tiles, margin=margin)       # host, whole-slide scan
    fast = T.PredictTileIndex(dev, tiling, margin=margin)                                       # device, binned
    assert len(fast) == len(slow)
    for i in range(len(slow)):
        a, b = slow[i], fast[i]
        for nt in ("tx", "bd"):
            assert set(a[nt].keys()) == set(b[nt].keys())
            for k, v in a[nt].items():
                if isinstance(v, torch.Tensor):
                    assert b[nt][k].is_cuda and torch.equal(v, b[nt][k].cpu()), (i, nt, k)
        for et in a.edge_types:
            assert torch.equal(a[et].edge_index, b[et].edge_index.cpu()), (i, et)
    assert all(bool((v == -1).all()) for v in fast._new_id.values())       # scratch map restored


def test_sampler_over_device_partition_equals_host(graphs):
    host, dev = graphs
    pd = T.partition_by_tiling(dev, T.SquareTiling(all_pos(dev), 25.0), margin=2.0)
    ph = T.partition_by_tiling(host, T.SquareTiling(all_pos(host), 25.0), margin=2.0)
    assert pd.weights("edge") == ph.weights("edge") and pd.weights("node") == ph.weights("node")
    cap = max(ph.weights("edge")) * 3
    for kw in (dict(), dict(shuffle=True, seed=5)):
        sd, sh = (T.TileBatchSampler(p, cap, mode="edge", skip_too_big=True, **kw) for p in (pd, ph))
        assert list(sd) == list(sh) and list(sd) == list(sh)               # (second pass: the re-packed epoch)


def test_slide_csr_slices_equal_host_stable_sort(graphs):
    """``TilePartition.build_csr`` (one HIP radix sort per edge store per slide, sliced per batch) against a stable
    argsort of the batch's own COO list on the host: single tiles, multi-tile batches, all three edge types."""
    from segger_amd.graph import batch_cache, edge_graph
    host, dev = graphs
    part = T.partition_by_tiling(dev, T.SquareTiling(all_pos(dev), 25.0), margin=2.0)
    part.build_csr()
    part.csr_max_tiles = 64
    n_t = len(part)
    for ids in ([0], [n_t - 1], [3, 1, 7], list(range(n_t))):
        b = part.batch(ids)
        for et in ETS:
            ei = b[et].edge_index
            ns, nd = b[et[0]].num_nodes, b[et[2]].num_nodes
            g = edge_graph(batch_cache(b), et, ei, ns, nd)
            eic = ei.cpu().long()
            for view, rows, cols, n_rows in ((g.by_dst, eic[1], eic[0], nd), (g.by_src, eic[0], eic[1], ns)):
                indptr, col, eid = naive_csr(rows, cols, n_rows)
                assert torch.equal(view.indptr.cpu(), indptr), (ids, et)
                assert torch.equal(view.col.cpu(), col) and torch.equal(view.eid.cpu(), eid), (ids, et)
            # a consumer that needs the by-source view the factory skipped ("lazy" + unique sources) can still get it
            lazy = edge_graph(batch_cache(b), et, ei, ns, nd, need_by_src="lazy")
            by_src = lazy.require_by_src()
            indptr, col, eid = naive_csr(eic[0], eic[1], ns)
            assert torch.equal(by_src.indptr.cpu(), indptr) and torch.equal(by_src.col.cpu(), col)


def test_slide_csr_sorted_in_tile_ranges_equals_one_sort(graphs):
    """Edge stores of 2^31 edges and more are sorted in ranges of whole tiles (reference _patches.py:1-9 hazard):
    with the limit lowered so that every store needs several sorts, the slide-level views are unchanged."""
    host, dev = graphs
    mk = lambda: T.partition_by_tiling(dev, T.SquareTiling(all_pos(dev), 25.0), margin=2.0)
    one, many = mk(), mk()
    one.build_csr()
    many.csr_sort_max_edges = max(int(many.edge_sizes[TX_TX].max()), 1) + 7
    assert len(many._sort_chunks(TX_TX)) > 3
    many.build_csr()
    for et in ETS:
        for side in ("by_dst", "by_src"):
            for k in ("ptr", "col", "eid"):
                assert torch.equal(one._csr[et][side][k], many._csr[et][side][k]), (et, side, k)
    many.csr_sort_max_edges = 1
    with pytest.raises(ValueError):
        many.build_csr()


def test_shard_of_a_csr_partition_hands_out_the_same_views(graphs):
    """``TilePartition.shard`` (a data-parallel rank's resident subset) carries the slide-level CSR slices along: batches
    of the shard get the same sorted views as the same batches of the full partition."""
    from segger_amd.graph import batch_cache, edge_graph
    host, dev = graphs
    part = T.partition_by_tiling(dev, T.SquareTiling(all_pos(dev), 25.0), margin=2.0)
    part.build_csr()
    part.csr_max_tiles = 8
    mine = list(range(1, len(part), 2))
    local = part.shard(mine)
    assert local.resident_bytes() < 0.7 * part.resident_bytes()
    for loc in ([0], [2, 1], [len(mine) - 1]):
        a, b = local.batch(loc), part.batch([mine[i] for i in loc])
        for et in ETS:
            assert torch.equal(a[et].edge_index, b[et].edge_index)
            ga = edge_graph(batch_cache(a), et, a[et].edge_index, a[et[0]].num_nodes, a[et[2]].num_nodes)
            gb = edge_graph(batch_cache(b), et, b[et].edge_index, b[et[0]].num_nodes, b[et[2]].num_nodes)
            for side in ("by_dst", "by_src"):
                x, y = getattr(ga, side), getattr(gb, side)
                assert torch.equal(x.indptr, y.indptr) and torch.equal(x.col, y.col) and torch.equal(x.eid, y.eid)


def test_staged_padded_batch_equals_partition_batch(cuda):
    """The captured training step's static buffers after ONE ``segger_stage`` launch: the real prefix of every array is
    the batch ``TilePartition.batch`` assembled (node features, the three CSR views == host stable sort of the batch's
    COO lists, rows-by-gene grouping, segmentation triplets), the padding is what train_step_graph.py promises (dummy
    nodes = copies of node 0, tx-tx pads = self-loops on dummies, tx-bd pads = dummy -> dummy, padded triplets = -1)
    and every padded view is a valid CSR of the padded sizes."""
    import math
    from segger_amd import LitISTEncoder
    from segger_amd.synthetic import make_fov
    from segger_amd.train_step_graph import GraphedTrainStep, step_bucket
    spec = SyntheticSpec(n_tx=40_000, n_bd=400, k_tx=8, seed=3)
    data, aux = make_fov(spec, cuda, return_aux=True)
    tiling = T.SquareTiling(data["tx"]["pos"], 10.0 * math.sqrt(spec.n_bd) / 3.0)
    part = T.partition_by_tiling(data, tiling, margin=5.0)
    part.build_csr()
    torch.manual_seed(0)
    m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
    m.model._materialize_bd(spec.bd_dim, "cpu")
    m.model.compute_dtype = torch.bfloat16
    m = m.to(cuda)
    m.set_similarities(aux["tx_similarity"].to(cuda), aux["bd_similarity"].to(cuda))
    m._max_epochs_override, m.current_epoch = 20, 10
    opt = m.configure_optimizers(capturable=True)
    for ids in ([4], [0, 5]):
        b = part.batch(ids)
        st = GraphedTrainStep(m, opt, step_bucket(b), b)
        st.stage(b)
        torch.cuda.synchronize()
        bc = b.to("cpu")
        n_tx, n_bd = bc["tx"].num_nodes, bc["bd"].num_nodes
        NT, NB = st.sizes["tx"], st.sizes["bd"]
        assert NT > n_tx and NB > n_bd
        # nodes
        for nt, n in (("tx", n_tx), ("bd", n_bd)):
            for a in ("x", "pos", "batch"):
                buf = st.nodes[nt][a].cpu()
                ref = bc[nt][a]
                # (integer stores are narrowed / widened losslessly; boundary features are staged in the compute dtype)
                same = torch.equal(buf[:n], ref.to(buf.dtype)) if buf.is_floating_point() else torch.equal(buf[:n].to(ref.dtype), ref)
                assert same, (nt, a)
                if a != "x" or nt == "bd":
                    assert bool((buf[n:] == buf[0]).all()), (nt, a)              # dummies: copies of node 0
        assert bool((st.nodes["tx"]["x"].cpu()[n_tx:] == spec.n_genes - 1).all())       # ... carrying the last gene id
        # CSR views: real part == host stable sort of the batch's COO list; the whole padded view is a valid CSR
        for et, graph, views in ((TX_TX, st.g_tt, ("by_dst", "by_src")), (TX_BD, st.g_tb, ("by_dst",))):
            eic = bc[et].edge_index.long()
            e = eic.shape[1]
            for side in views:
                v = getattr(graph, side)
                rows, cols = (eic[1], eic[0]) if side == "by_dst" else (eic[0], eic[1])
                n_rows_real = (n_bd if et == TX_BD else n_tx) if side == "by_dst" else n_tx
                n_cols_real = n_tx
                indptr, col, eid = naive_csr(rows, cols, n_rows_real)
                ip, cc, ee = v.indptr.cpu(), v.col.cpu(), v.eid.cpu()
                assert torch.equal(ip[: n_rows_real + 1], indptr), (et, side)
                assert torch.equal(cc[:e], col) and torch.equal(ee[:e], eid), (et, side)
                assert int(ip[-1]) == v.n_edges == cc.numel() and bool((ip[1:] >= ip[:-1]).all())
                assert torch.equal(ee[e:].long(), torch.arange(e, v.n_edges))     # padding edges continue the ids
                row_of = torch.repeat_interleave(torch.arange(v.n_rows), ip[1:] - ip[:-1])
                assert bool((row_of[e:] >= n_rows_real).all()) and bool((cc[e:].long() >= n_cols_real).all())
                assert bool((cc.long() < v.n_cols).all()) and bool((cc >= 0).all())
                if et == TX_TX:
                    assert torch.equal(cc[e:].long(), row_of[e:])                 # self-loops on the dummies
            if et == TX_TX:                                                       # identical padding in both views
                assert torch.equal(graph.by_dst.col.cpu()[e:], graph.by_src.col.cpu()[e:])
        # rows-by-gene grouping: a valid grouping of ALL padded rows whose real part is the batch's
        ids_pad = st.nodes["tx"]["x"].cpu().long()
        gi, gc = st.by_gene.indptr.cpu(), st.by_gene.col.cpu().long()
        assert int(gi[-1]) == NT and torch.equal(torch.sort(gc).values, torch.arange(NT))
        gene_of_slot = torch.repeat_interleave(torch.arange(spec.n_genes), gi[1:] - gi[:-1])
        assert torch.equal(ids_pad[gc], gene_of_slot)
        # segmentation triplets
        e_tb = bc[TX_BD].edge_index.shape[1]
        assert torch.equal(st.sg_src.cpu()[:e_tb], bc[TX_BD].edge_index[0].long())
        assert torch.equal(st.sg_pos.cpu()[:e_tb], bc[TX_BD].edge_index[1].long())
        assert bool((st.sg_pos.cpu()[e_tb:] == -1).all()) and bool((st.sg_src.cpu()[e_tb:] >= n_tx).all())
        assert int(st.n_bd.cpu()) == n_bd


def test_generic_geometry_trains_on_csr_partition(cuda):
    """A (heads, channels) without a specialised kernel needs the by-source view the slide-level factory skips for
    unique-source edge stores: the EdgeGraph now carries the batch's COO list and sorts on demand (round-2 advice)."""
    import math
    from segger_amd import LitISTEncoder
    from segger_amd.synthetic import make_fov
    spec = SyntheticSpec(n_tx=20_000, n_bd=200, k_tx=6, seed=2)
    data, aux = make_fov(spec, cuda, return_aux=True)
    part = T.partition_by_tiling(data, T.SquareTiling(data["tx"]["pos"], 10.0 * math.sqrt(spec.n_bd) / 2.0), margin=5.0)
    part.build_csr()
    torch.manual_seed(0)
    m = LitISTEncoder(n_genes=spec.n_genes, in_channels=32, hidden_channels=24, out_channels=24, n_heads=2)
    m.model._materialize_bd(spec.bd_dim, "cpu")
    m = m.to(cuda)
    m.set_similarities(aux["tx_similarity"].to(cuda), aux["bd_similarity"].to(cuda))
    m._max_epochs_override, m.current_epoch = 20, 10
    m.train()
    loss = m.training_step(part.batch([1]), 0)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_rank_local_fov_generation_equals_shards_of_the_whole_fov(cuda):
    """``fov.build_fov_shard`` (a data-parallel rank builds only the edges of its own tiles; nodes and a chunked counting
    pass are replicated) against ``build_fov_batches`` + ``dp.rank_schedule`` + ``TilePartition.shard`` on the whole FOV:
    the same batch list and schedule on every rank, and tile for tile the same tensors -- node stores, tile-local edge
    lists, global permutations, pointer arrays, slide-level sorted views -- bit for bit; its peak memory stays below the
    whole-FOV route's."""
    from segger_amd.dp import rank_schedule
    from segger_amd.fov import batch_weights, build_fov_batches, build_fov_shard
    spec = SyntheticSpec(n_tx=300_000, n_bd=3_000, k_tx=15, seed=11)
    kw = dict(tile_nodes=20_000, edges_per_batch=700_000)
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    part, batches, aux, tiling = build_fov_batches(spec, cuda, **kw)
    peak_full = torch.cuda.max_memory_allocated() - base
    weights = batch_weights(part, batches)
    world = 3
    sched = rank_schedule(weights, world)
    assert len(batches) >= 4 and len(part) >= 9
    seen = set()
    for rank in range(world):
        mine = [k for k in sched[rank] if k is not None]
        tiles = sorted({t for k in mine for t in batches[k]})
        ref = part.shard(tiles)
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loc, b2, local_batches, s2, aux2, tiling2, info = build_fov_shard(spec, cuda, rank, world, count_chunk=70_001, **kw)
        peak_local = torch.cuda.max_memory_allocated() - base
        assert b2 == batches and s2 == sched and info["tiles"] == tiles and info["weights"] == weights
        assert sorted(local_batches) == sorted(mine)
        assert len(tiling2) == len(tiling) and torch.equal(aux2["tx_similarity"], aux["tx_similarity"])
        assert loc.num_tiles == ref.num_tiles == len(tiles)
        for nt in ("tx", "bd"):
            assert torch.equal(loc.node_indptr[nt], ref.node_indptr[nt]) and torch.equal(loc.node_perm[nt], ref.node_perm[nt])
            assert set(loc.data[nt].keys()) == set(ref.data[nt].keys())
            for a, v in ref.data[nt].items():
                if isinstance(v, torch.Tensor):
                    assert torch.equal(loc.data[nt][a], v), (nt, a)
        for et in ETS:
            assert torch.equal(loc.edge_indptr[et], ref.edge_indptr[et]), et
            assert torch.equal(loc.data[et]["edge_index"], ref.data[et]["edge_index"]), et
            for side in ("by_dst", "by_src"):
                for k in ("ptr", "col", "eid"):
                    assert torch.equal(loc._csr[et][side][k], ref._csr[et][side][k]), (et, side, k)
        for k in mine:                                       # and the batches the rank trains on
            a, b = loc.batch(local_batches[k]), part.batch(batches[k])
            assert torch.equal(a["tx"]["pos"], b["tx"]["pos"]) and torch.equal(a[TX_TX].edge_index, b[TX_TX].edge_index)
            assert info["units"][k] == (int(sum(part.edge_sizes[TX_BD][t] for t in batches[k])),
                                        int(sum(part.edge_sizes[TX_TX][t] for t in batches[k])))
        assert peak_local < 0.8 * peak_full, (peak_local, peak_full)
        seen |= set(tiles)
        del loc, ref
    assert seen == {t for ids in batches for t in ids}
