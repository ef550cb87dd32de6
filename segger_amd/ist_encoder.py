"""MI355X-native ``ISTEncoder``: same module tree, parameter names and forward
contract as reference ``src/segger/models/ist_encoder.py`` (so reference
state dicts load), with the PyG ``HeteroConv``/``GATv2Conv`` message passing
replaced by the fused HIP kernels of ``libsegger_amd.so``.

Reference map
-------------
``sinusoidal_embedding``      ist_encoder.py:22-31
``Positional2dEmbedder``      ist_encoder.py:33-79
``SkipGAT``                   ist_encoder.py:82-211   (HeteroConv of GATv2Conv, dropout 0.2)
``ISTEncoder``                ist_encoder.py:214-333

Differences that are deliberate: the never-used
``('bd','contains','tx')`` conv is not built (it has no edges in segger's data
and its lazy parameters are never materialised); the per-graph min/max loop of
the positional embedder (one host sync per graph, ``:69-73``) is a single
segmented reduction.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from torch.nn import Embedding, Linear, Module, ModuleDict, ModuleList, Parameter, Sequential, SiLU
from torch.nn import functional as F

from . import ops
from .graph import EdgeGraph, edge_graph
from .hetero import TX_BD, TX_TX, EdgeType

GAT_DROPOUT = 0.2          # ist_encoder.py:116,123
NEGATIVE_SLOPE = 0.2       # GATv2Conv default


def _layer_seed(seed, which: int):
    """(2*layer + edge-type slot) with the optional device counter carried along; a third entry is a constant added to the
    seed (a captured step reads the counter BEFORE its end-of-step advance: offset = the advance)."""
    if isinstance(seed, tuple):
        return (2 * int(seed[0]) + which + (int(seed[2]) if len(seed) > 2 else 0), seed[1])
    return 2 * int(seed) + which


def pyg_key(edge_type: EdgeType) -> str:
    """torch_geometric's ModuleDict key for a tuple (state-dict compatibility)."""
    return "<" + "___".join(edge_type) + ">"


def sinusoidal_embedding(x: Tensor, dim: int, max_period: float = 1000) -> Tensor:
    """cos|sin features of a flat tensor (ist_encoder.py:22-31); fp32 like the reference."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=x.device) / half)
    args = x[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


FUSED_POSMLP = True      # default of Positional2dEmbedder.fused (tools flip it for A/B runs)


class Positional2dEmbedder(Module):
    """Per-graph min-max normalised (x, y) -> 2 x sinusoid(256) -> shared MLP -> concat."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256):
        super().__init__()
        self.dim = hidden_size // 2
        self.mlp = Sequential(
            Linear(frequency_embedding_size, self.dim, bias=True),
            SiLU(),
            Linear(self.dim, self.dim, bias=True),
        )
        self.frequency_embedding_size = frequency_embedding_size
        self.fused = FUSED_POSMLP    # one-kernel route (ops.posmlp) where it applies; False: posfreq + linear + SiLU + linear

    @staticmethod
    def normalize(pos: Tensor, batch: Optional[Tensor], num_graphs: Optional[int] = None) -> Tensor:
        pos = pos.float()
        if batch is None:                                   # ist_encoder.py:62-64 (no epsilon)
            pos = pos - pos.min(dim=0).values
            return pos / pos.max(dim=0).values
        if num_graphs is None:
            num_graphs = int(batch.max()) + 1 if batch.numel() else 0
        mins, maxs = ops.segment_minmax(pos, batch, num_graphs)     # one HIP pass, no per-graph sync
        lo, hi = mins[batch.long()], maxs[batch.long()]
        return (pos - lo) / (hi - lo + 1e-8)                # ist_encoder.py:74

    def forward(self, pos: Tensor, batch: Optional[Tensor] = None, *, num_graphs: Optional[int] = None,
                dtype: torch.dtype = torch.float32, gelu: bool = False, return_pre: bool = False, minmax=None):
        """``gelu`` (not in the reference): also apply the GELU that ISTEncoder puts on its concatenated input;
        ``return_pre`` (with ``gelu``): ``(gelu(h), h)`` as ``ops.posmlp`` documents it, or ``(gelu(h), None)`` on the
        routes that keep the GELU in autograd."""
        n = pos.shape[0]
        fd = self.frequency_embedding_size
        if (batch is not None or pos.is_cuda) and fd % 16 == 0:
            # fused: per-graph min/max (one pass) -> normalise + sinusoid written straight in `dtype`
            # (no batch vector = one graph, and the reference normalises it WITHOUT the epsilon, ist_encoder.py:62-64)
            eps_n = 1e-8 if batch is not None else 0.0
            if batch is None:
                num_graphs = 1
            elif num_graphs is None:
                num_graphs = int(batch.max()) + 1 if batch.numel() else 0
            l0, l2 = self.mlp[0], self.mlp[2]
            use_fused = (self.fused and pos.is_cuda and l0.bias is not None and l2.bias is not None
                         and ops.posmlp_supported(fd, self.dim, dtype))
            # (the fused kernel and posfreq only look up the graphs of existing nodes: no (0, 0) fix-up for empty ones)
            mins, maxs = ops.segment_minmax(pos, batch, num_graphs, keep_empty=True, out=minmax)
            if use_fused:
                # sinusoid + Linear + SiLU + Linear in one kernel: the [2n, 256] feature matrix is generated in
                # registers (and stored once for the weight gradient when training) instead of written and re-read
                return ops.posmlp(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, dtype,
                                  eps=eps_n, max_period=10000.0, gelu=gelu, return_pre=return_pre and gelu)
            if (dtype == torch.float32 and ops.F32_GATE_EPILOGUE and l0.bias is not None and l2.bias is not None
                    and ops.pos_poly_mlp_f32_supported(pos, l0.weight, l2.weight)):
                # fp32 storage: the first Linear as a polynomial of the normalised coordinate (no feature matrix, no K = 256
                # GEMM: csrc/posenc_poly.hip), SiLU and the 64-wide second Linear (+ the GELU that follows) behind one node
                if gelu and return_pre:
                    h, gh = ops.pos_poly_mlp_f32(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, eps=eps_n,
                                                 max_period=10000.0, gelu_out=True)
                    return gh.reshape(n, -1), h.reshape(n, -1)
                h = ops.pos_poly_mlp_f32(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, eps=eps_n,
                                         max_period=10000.0).reshape(n, -1)
                return F.gelu(h) if gelu else h
            freq = ops.posfreq(pos, batch, mins, maxs, fd, dtype, eps=eps_n, max_period=10000.0)
        else:
            pos = self.normalize(pos, batch, num_graphs)
            freq = sinusoidal_embedding(pos.flatten(), fd, max_period=10000).reshape(n, 2, fd).to(dtype)
        l0, l2 = self.mlp[0], self.mlp[2]
        flat = freq.reshape(-1, fd)
        if (dtype == torch.float32 and ops.F32_GATE_EPILOGUE and l0.bias is not None and l2.bias is not None
                and ops.mlp_silu_f32_supported(flat, l0.weight, l2.weight)):
            # fp32 storage: the MLP as one autograd node (SiLU, and the GELU that follows, in the GEMMs' epilogues; SiLU's
            # derivative in the data-gradient GEMM's epilogue)
            if gelu and return_pre:
                h, gh = ops.mlp_silu_f32(flat, l0.weight, l0.bias, l2.weight, l2.bias, gelu_out=True)
                return gh.reshape(n, -1), h.reshape(n, -1)
            h = ops.mlp_silu_f32(flat, l0.weight, l0.bias, l2.weight, l2.bias).reshape(n, -1)
        else:
            h = F.silu(ops.linear(freq, l0.weight, l0.bias))
            h = ops.linear(h, l2.weight, l2.bias).flatten(-2)
        if gelu and return_pre:
            if dtype == torch.float32 and h.is_cuda and ops.F32_GATE_EPILOGUE:
                # as ops.posmlp documents it: gelu(h) is a constant for autograd and the gradient arrives for h -- the consumer
                # (ops.embed_linear) applies gelu'(h) in the epilogue of its data-gradient GEMM instead of a gelu_backward pass
                return F.gelu(h).detach(), h
            return F.gelu(h), None
        return F.gelu(h) if gelu else h


def _pair_node(emb: "Positional2dEmbedder", pos_a, batch_a, pos_b, batch_b, num_graphs, dtype):
    """``((gelu(pe_a), pe_a), pe_b)`` from ONE autograd node (``ops.posmlp_pair``), or None where that route does not
    apply -- then the caller embeds each node type by its own call."""
    fd = emb.frequency_embedding_size
    l0, l2 = emb.mlp[0], emb.mlp[2]
    if not (POS_PAIR_NODE and emb.fused and pos_a.is_cuda and fd % 16 == 0 and batch_a is not None and batch_b is not None
            and num_graphs is not None and ops.posmlp_pair_supported(l0.weight, l0.bias, l2.weight, l2.bias, dtype)):
        return None
    mm_a = ops.segment_minmax(pos_a, batch_a, num_graphs, keep_empty=True)
    mm_b = ops.segment_minmax(pos_b, batch_b, num_graphs, keep_empty=True)
    return ops.posmlp_pair(pos_a, batch_a, mm_a[0], mm_a[1], pos_b, batch_b, mm_b[0], mm_b[1], l0.weight, l0.bias, l2.weight,
                           l2.bias, dtype, eps=1e-8, max_period=10000.0)


class _SplitRows(torch.autograd.Function):
    """[n_a + n_b, D] -> ([n_a, D], [n_b, D]) as views; backward = one ``cat`` (autograd's own slicing would zero-fill
    and add a full-size matrix per slice)."""

    @staticmethod
    def forward(ctx, x, n_a):
        ctx.n = (int(n_a), int(x.shape[0]) - int(n_a))
        return x[:n_a], x[n_a:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros((ctx.n[0],) + tuple(ref.shape[1:]))
        if gb is None:
            gb = ref.new_zeros((ctx.n[1],) + tuple(ref.shape[1:]))
        return torch.cat((ga, gb), 0), None


MERGED_POS_EMBED = True     # one embedder call for both node types (tools flip it for A/B runs)
FRONT_JOIN = True           # ... and one gather / concat / GELU launch for both (ops.front_join); False: per type, torch on 'bd'
POS_PAIR_NODE = True        # large batches (one embedder call per type): both calls behind one autograd node (ops.posmlp_pair)


class GATv2Conv(Module):
    """Parameter holder with torch_geometric.nn.GATv2Conv's names and shapes
    (``lin_l``, ``lin_r``: [H*C, in] + bias; ``att``: [1, H, C]; ``bias``: [H*C]).
    The arithmetic lives in :func:`segger_amd.ops.gatv2_aggregate`."""

    def __init__(self, in_channels: Tuple[int, int], out_channels: int, heads: int,
                 negative_slope: float = NEGATIVE_SLOPE, dropout: float = GAT_DROPOUT):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope, self.dropout = negative_slope, dropout
        self.lin_l = Linear(in_channels[0], heads * out_channels, bias=True)
        self.lin_r = Linear(in_channels[1], heads * out_channels, bias=True)
        self.att = Parameter(torch.empty(1, heads, out_channels))
        self.bias = Parameter(torch.empty(heads * out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        # PyG: glorot for lin weights and att, zeros for biases
        for lin in (self.lin_l, self.lin_r):
            torch.nn.init.xavier_uniform_(lin.weight)
            torch.nn.init.zeros_(lin.bias)
        a = math.sqrt(6.0 / (self.att.size(-2) + self.att.size(-1)))
        torch.nn.init.uniform_(self.att, -a, a)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x: Tuple[Tensor, Tensor], graph: EdgeGraph, *, apply_gelu: bool = False, seed: int = 0,
                return_attention_weights: bool = False):
        x_src, x_dst = x
        dt = x_src.dtype
        xl = ops.linear(x_src, self.lin_l.weight, self.lin_l.bias)
        xr = ops.linear(x_dst, self.lin_r.weight, self.lin_r.bias)
        p = self.dropout if self.training else 0.0
        return ops.gatv2_aggregate(xl, xr, self.att, self.bias, graph, self.heads, self.out_channels,
                                   apply_gelu=apply_gelu, negative_slope=self.negative_slope,
                                   dropout_p=p, seed=seed, return_alpha=return_attention_weights)


class _HeteroConv(Module):
    """Holds ``convs`` under PyG's mangled keys so ``state_dict`` matches HeteroConv."""

    def __init__(self, convs: Dict[EdgeType, Module]):
        super().__init__()
        self.convs = ModuleDict({pyg_key(k): v for k, v in convs.items()})

    def __getitem__(self, et: EdgeType) -> Module:
        return self.convs[pyg_key(et)]


class SkipGAT(Module):
    """One hetero GATv2 layer over tx-neighbors-tx and tx-belongs-bd
    (ist_encoder.py:82-211) + the GELU that follows it in ISTEncoder (``:325``),
    fused.  ``attention_weights`` holds tx-neighbors-tx coefficients of the last
    forward when ``store_attention`` is set (the reference's forward hook)."""

    def __init__(self, in_channels: Tuple[int, int], out_channels: int, n_heads: int,
                 add_self_loops_tx: bool = False):
        super().__init__()
        if add_self_loops_tx:
            raise NotImplementedError("add_self_loops_tx=True is never used by segger (ist_encoder.py:104)")
        self.out_channels, self.n_heads = out_channels, n_heads
        self.conv = _HeteroConv({
            TX_TX: GATv2Conv((in_channels[0], in_channels[0]), out_channels, n_heads),
            TX_BD: GATv2Conv((in_channels[0], in_channels[1]), out_channels, n_heads),
        })
        self.store_attention = False
        self._attn_weights: Dict[EdgeType, Tensor] = {}

    def forward(self, x_dict: Dict[str, Tensor], edge_index_dict: Dict[EdgeType, Tensor], *,
                graphs: Optional[Dict[EdgeType, EdgeGraph]] = None, apply_gelu: bool = False,
                seed=0, keep_bits: Optional[Dict[EdgeType, tuple]] = None) -> Dict[str, Tensor]:
        for et in (TX_TX, TX_BD):
            if et not in edge_index_dict:
                raise KeyError(f"edge type {et} missing from edge_index_dict: segger's HeteroConv would "
                               f"drop node type '{et[2]}' and fail in the next layer")
        x_tx, x_bd = x_dict["tx"], x_dict["bd"]
        if graphs is None:
            graphs = {et: edge_graph(None, et, edge_index_dict[et], x_dict[et[0]].shape[0], x_dict[et[2]].shape[0],
                                     validate="deferred")
                      for et in (TX_TX, TX_BD)}
        tt, tb = self.conv[TX_TX], self.conv[TX_BD]
        dt = x_tx.dtype
        # one fused projection for the three linear maps that read x_tx (stacked and cast once per optimizer step)
        w_tx, b_tx = (tt.lin_l.weight, tt.lin_r.weight, tb.lin_l.weight), (tt.lin_l.bias, tt.lin_r.bias, tb.lin_l.bias)
        if isinstance(x_tx, ops.EmbedInput):
            xp_tx = ops.embed_linear(x_tx, w_tx, b_tx)
            xp_bd = ops.linear(x_bd, tb.lin_r.weight, tb.lin_r.bias)
        else:                    # both node types' projections in one launch (the boundary side rides in the grid)
            xp_tx, xp_bd = ops.linear_pair(x_tx, w_tx, b_tx, x_bd, tb.lin_r.weight, tb.lin_r.bias)
        p = tt.dropout if self.training else 0.0
        y_tx, y_bd, alpha = ops.hetero_gat_layer(
            xp_tx, xp_bd, tt.att, tt.bias, tb.att, tb.bias, graphs[TX_TX], graphs[TX_BD],
            self.n_heads, self.out_channels, apply_gelu=apply_gelu, negative_slope=tt.negative_slope,
            dropout_p=p, seed_tt=_layer_seed(seed, 0), seed_tb=_layer_seed(seed, 1), return_alpha=self.store_attention,
            bits_tt=None if keep_bits is None else keep_bits.get(TX_TX),
            bits_tb=None if keep_bits is None else keep_bits.get(TX_BD))
        if self.store_attention:
            self._attn_weights[TX_TX] = alpha
        return {"tx": y_tx, "bd": y_bd}

    @property
    def attention_weights(self) -> Dict[EdgeType, Tensor]:
        if not self._attn_weights:
            raise AttributeError("Attention weights are empty. Please perform a forward pass.")
        return self._attn_weights


class _HeteroDictLinear(Module):
    """``lins.{type}`` naming of torch_geometric.nn.HeteroDictLinear."""

    def __init__(self, in_channels: int, out_channels: int, types=("tx", "bd")):
        super().__init__()
        self.lins = ModuleDict({t: Linear(in_channels, out_channels, bias=True) for t in types})

    def forward(self, x_dict: Dict[str, Tensor]) -> Dict[str, Tensor]:
        out = {}
        if len(x_dict) == 2:     # lin_last of both node types: one launch
            (ka, xa), (kb, xb) = x_dict.items()
            la, lb = self.lins[ka], self.lins[kb]
            out[ka], out[kb] = ops.linear_pair(xa, la.weight, la.bias, xb, lb.weight, lb.bias)
            return out
        for k, x in x_dict.items():
            lin = self.lins[k]
            out[k] = ops.linear(x, lin.weight, lin.bias)
        return out


class ISTEncoder(Module):
    """Same constructor and ``forward(x_dict, edge_index_dict, pos_dict, batch_dict)``
    as the reference (ist_encoder.py:219-333).

    Extra keyword-only knobs (not in the reference): ``bd_in_channels`` (the
    reference's lazy ``Linear(-1, in)`` is materialised on first forward when
    this is None), ``compute_dtype`` (activations; parameters stay fp32).
    """

    def __init__(self, n_genes: int, in_channels: int = 16, hidden_channels: int = 32, out_channels: int = 32,
                 n_mid_layers: int = 3, n_heads: int = 3, normalize_embeddings: bool = True,
                 use_positional_embeddings: bool = True, *, bd_in_channels: Optional[int] = None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.normalize_embeddings = normalize_embeddings
        self.use_positional_embeddings = use_positional_embeddings
        self.compute_dtype = compute_dtype
        self.hparams = dict(n_genes=n_genes, in_channels=in_channels, hidden_channels=hidden_channels,
                            out_channels=out_channels, n_mid_layers=n_mid_layers, n_heads=n_heads,
                            normalize_embeddings=normalize_embeddings,
                            use_positional_embeddings=use_positional_embeddings)
        self.in_channels, self.n_heads = in_channels, n_heads
        # 16-bit compute: first-layer projections as per-gene table + positional GEMM (ops.embed_linear); its ~15 extra
        # tiny launches (table GEMM, weight slices) only pay for themselves on large batches
        self.split_first_layer = True
        self.split_first_layer_min_rows = 200_000
        # fp32 storage: the un-split first layer is a K = 256 GEMM on the exact-fp32 MFMA pipe (157 TFLOP/s) forward, backward
        # and for its weight gradient -- 0.6 ms of the captured 1M-edge step's 2.4 (profiles/r06_small_batch_step_f32_*.txt) --
        # while the split form's K = 128 shapes run on the bf16x3 kernels: worth it from far fewer rows
        self.split_first_layer_min_rows_f32 = 4_096
        self.lin_first = ModuleDict({"tx": Embedding(n_genes, in_channels)})
        if bd_in_channels is not None:
            self.lin_first["bd"] = Linear(bd_in_channels, in_channels)
        self.pos_emb = Positional2dEmbedder(in_channels)
        f0 = 2 * in_channels if use_positional_embeddings else in_channels
        self.conv_layers = ModuleList()
        self.conv_layers.append(SkipGAT((f0, f0), hidden_channels, n_heads))
        for _ in range(n_mid_layers):
            self.conv_layers.append(SkipGAT((hidden_channels * n_heads,) * 2, hidden_channels, n_heads))
        last_in = hidden_channels * n_heads
        self.conv_layers.append(SkipGAT((last_in, last_in), out_channels, n_heads))
        self.lin_last = _HeteroDictLinear(out_channels * n_heads, out_channels, types=("tx", "bd"))
        # dropout stream: effective seed of (layer, edge type, step) = 2*layer + type + *_step_dev; the counter
        # lives on the device and advances by 256 per training forward, so captured graphs see fresh masks
        self.register_buffer("_step_dev", torch.zeros(1, dtype=torch.int64), persistent=False)
        self.register_load_state_dict_pre_hook(ISTEncoder._adopt_reference_state_dict)

    @staticmethod
    def _adopt_reference_state_dict(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                    error_msgs) -> None:
        """Makes a reference checkpoint load with ``strict=True``: the reference's lazy ``Linear(-1, in)`` for
        boundaries (ist_encoder.py:261) is materialised from the shape of the incoming ``lin_first.bd.weight``
        (instead of being fabricated at random by the first forward), and the entries of the never-initialised
        ``('bd','contains','tx')`` conv (lazy placeholders, no trained values) are dropped."""
        w = state_dict.get(prefix + "lin_first.bd.weight")
        if w is not None and "bd" not in module.lin_first and getattr(w, "dim", lambda: 0)() == 2:
            ref = module.lin_first["tx"].weight
            module.lin_first["bd"] = Linear(int(w.shape[1]), int(w.shape[0])).to(device=ref.device)
        for k in [k for k in state_dict if k.startswith(prefix) and "<bd___contains___tx>" in k]:
            del state_dict[k]

    def plane_views(self, graphs, seed_offset: int = 0):
        """The CSR views whose attention-dropout masks a training step needs as bit planes: [(edge type, side, csr, seeds)]
        (side 0 = by destination, 1 = by source), or None when the planes do not apply."""
        n_layers = len(self.conv_layers)
        first = self.conv_layers[0]
        if n_layers > 16 or self.n_heads > 8 or not (first.conv[TX_TX].dropout > 0):
            return None
        views = []
        for which, et in ((0, TX_TX), (1, TX_BD)):
            g = graphs.get(et)
            if g is None or g.by_dst is None:
                continue
            seeds = [2 * li + which + int(seed_offset) for li in range(n_layers)]
            views += [(et, 0, g.by_dst, seeds)] + ([(et, 1, g.by_src, seeds)] if g.by_src is not None else [])
        return views

    @staticmethod
    def planes_of(views, bits) -> dict:
        out = {}
        for (et, side, _, _), b in zip(views, bits):
            out.setdefault(et, [None, None])[side] = b
        return {et: tuple(v) for et, v in out.items()}

    def _dropout_planes(self, graphs, step):
        """The attention-dropout masks of all layers as bit planes per CSR view (``ops.dropout_bits_many``): one launch
        per step; the 12 aggregation launches of the step then test a bit per (edge, head) instead of hashing."""
        views = self.plane_views(graphs)
        if views is None:
            return None
        p = self.conv_layers[0].conv[TX_TX].dropout
        return self.planes_of(views, ops.dropout_bits_many([(c, sd) for _, _, c, sd in views], self.n_heads, p, step))

    def _pos_embed_pair(self, pos_dict, batch_dict, num_graphs, dt, gelu: bool, graphs, joint: bool = False,
                        bd_plain: bool = False):
        """(pe_tx, pe_bd, None, plain): ``pos_emb`` of both node types (``plain``: pe_bd comes without the GELU), in one call where the batch vectors allow it.  With ``gelu``
        pe_tx is the pair ``(gelu(h), h or None)`` of ``Positional2dEmbedder.forward(return_pre=True)``.  ``joint``: when
        the two types were embedded by one call (and ``gelu`` is off), return ``(None, None, pe)`` instead, ``pe`` the
        un-split [n_tx + n_bd, D] matrix (for :func:`ops.front_join`).  ``bd_plain``: with one call per type, the
        boundaries' embedding comes WITHOUT the GELU whatever ``gelu`` says (``ops.front_join`` applies it)."""
        b_tx, b_bd = batch_dict.get("tx"), batch_dict.get("bd")
        staged = graphs.get("pos_all") if graphs is not None else None
        # (large batches -- the `split` route -- keep one call per type: there the launches do not matter, and joining
        # the two gradients of the embedder's output would copy a [n_tx, D] matrix)
        if staged is None and (not MERGED_POS_EMBED or gelu or b_tx is None or b_bd is None or num_graphs is None):
            if gelu and bd_plain:
                # one call per type, ONE autograd node: the embedder's parameters receive one gradient (16-bit fused route)
                both = _pair_node(self.pos_emb, pos_dict["tx"], b_tx, pos_dict["bd"], b_bd, num_graphs, dt)
                if both is not None:
                    return both[0], both[1], None, True
            one = lambda k, g=gelu, **kw: self.pos_emb(pos_dict[k], batch_dict.get(k), num_graphs=num_graphs, dtype=dt,
                                                       gelu=g, **kw)
            return (one("tx", return_pre=True) if gelu else one("tx")), one("bd", gelu and not bd_plain), None, bd_plain or not gelu
        if staged is not None:                               # a captured step stages the concatenation itself
            pos_all, batch_all = staged
        else:
            pos_all = torch.cat((pos_dict["tx"].float(), pos_dict["bd"].float()), 0)
            batch_all = torch.cat((b_tx.long(), b_bd.long() + int(num_graphs)), 0)
        pe = self.pos_emb(pos_all, batch_all, num_graphs=2 * int(num_graphs), dtype=dt, gelu=gelu,
                          minmax=graphs.get("minmax") if graphs is not None else None)
        if joint and not gelu:
            return None, None, pe, True
        pe_tx, pe_bd = _SplitRows.apply(pe, int(pos_dict["tx"].shape[0]))
        return ((pe_tx, None) if gelu else pe_tx), pe_bd, None, not gelu

    def _materialize_bd(self, d_in: int, device) -> None:
        if "bd" not in self.lin_first:
            self.lin_first["bd"] = Linear(d_in, self.in_channels).to(device)

    def forward(self, x_dict: Dict[str, Tensor], edge_index_dict: Dict[EdgeType, Tensor],
                pos_dict: Dict[str, Tensor], batch_dict: Dict[str, Tensor], *,
                num_graphs: Optional[int] = None, cache: Optional[dict] = None,
                graphs: Optional[Dict[EdgeType, EdgeGraph]] = None) -> Dict[str, Tensor]:
        dt = self.compute_dtype
        self._materialize_bd(x_dict["bd"].shape[-1], x_dict["bd"].device)
        bd_lin = self.lin_first["bd"]
        emb = self.lin_first["tx"]
        x_bd = ops.linear(x_dict["bd"].to(dt), bd_lin.weight, bd_lin.bias)
        if self.use_positional_embeddings:
            fused_tx = self.in_channels % 32 == 0 and emb.weight.dtype == torch.float32
            split = False
            if fused_tx:
                first = self.conv_layers[0].conv
                m_first = sum(int(w.shape[0]) for w in (first[TX_TX].lin_l.weight, first[TX_TX].lin_r.weight,
                                                        first[TX_BD].lin_l.weight))
                probe = ops.EmbedInput(emb.weight, x_dict["tx"], x_bd[:0, : self.in_channels], None)
                split = (self.split_first_layer
                         and x_dict["tx"].shape[0] >= (self.split_first_layer_min_rows_f32 if dt == torch.float32
                                                       else self.split_first_layer_min_rows)
                         and ops.embed_linear_supported(probe, m_first))
            # ONE embedder call for both node types (the reference calls it per type, ist_encoder.py:314-318): graph ids
            # of the boundaries are offset by num_graphs, so the per-graph min / max stay per type.  Half the launches
            # of the front end, and the embedder's parameters receive ONE gradient each (what lets a captured step
            # postpone its partial sums, ops.deferred_reductions).  `split`: the GELU of ist_encoder.py:320 comes
            # applied (gelu(cat(a, b)) = cat(gelu(a), gelu(b))).
            join = fused_tx and FRONT_JOIN
            pe_tx, pe_bd, pe_all, bd_plain = self._pos_embed_pair(pos_dict, batch_dict, num_graphs, dt, split, graphs,
                                                                  joint=join, bd_plain=join)
            pre_tx = None
            if split:
                pe_tx, pre_tx = pe_tx
            if pe_all is None and join and bd_plain and pe_bd.dtype == x_bd.dtype:
                # one embedder call per type: the boundary side alone through the join (no transcript rows, no table gradient)
                _, x_bd = ops.front_join(emb.weight.detach(), x_dict["tx"][:0], x_bd, pe_bd, None)
            elif pe_all is None:
                # `bd_plain`: pe_bd came back WITHOUT its GELU (the join applies it), whatever `split` says for the
                # transcripts -- keyed on the flag the embedder returned, not on `split` (ist_encoder.py:320)
                gelu_on_pe = bd_plain or not split
                x_bd = F.gelu(torch.cat((x_bd, pe_bd), -1)) if gelu_on_pe else torch.cat((F.gelu(x_bd), pe_bd), -1)
            if fused_tx:
                # gather + concat + GELU in one kernel; its table gradient sums over rows grouped by gene id: one
                # sort per batch (not needed without grad), cached with the batch or supplied with `graphs`
                ids = x_dict["tx"]
                by_gene = graphs.get("tx_by_gene") if graphs is not None else None
                if by_gene is None and torch.is_grad_enabled() and emb.weight.requires_grad:
                    # batches of a resident partition keep what depends on the tile set only across epochs
                    # (tiles.TilePartition: cache["persistent"], shared with the captured step's staging)
                    store = cache.get("persistent") if cache is not None else None
                    if store is not None:
                        by_gene = store.get("tx_by_gene")
                        if by_gene is None or by_gene.n_rows != emb.weight.shape[0] or by_gene.n_edges != ids.shape[0]:
                            by_gene = store["tx_by_gene"] = ops.rows_by_id(ids, emb.weight.shape[0])
                    else:
                        key = ("by_gene", ids.data_ptr(), int(ids.shape[0]))
                        by_gene = cache.get(key) if cache is not None else None
                        if by_gene is None:
                            by_gene = ops.rows_by_id(ids, emb.weight.shape[0])
                            if cache is not None:
                                cache[key] = by_gene
                if pe_all is not None:
                    # both node types: gather / concat / GELU in one launch, and back in one (the boundary side's torch cat +
                    # GELU + their backward, and the full-size cat joining the two slices' gradients of the embedder's output)
                    x_tx, x_bd = ops.front_join(emb.weight, ids, x_bd, pe_all, by_gene)
                elif split:
                    # keep gelu(cat(E[g], pe)) as its parts: the first layer projects it as T[g] + W_pe gelu(pe)
                    x_tx = ops.EmbedInput(emb.weight, ids.to(torch.int32).contiguous(), pe_tx, by_gene, pre_tx)
                else:
                    x_tx = ops.embed_gelu(emb.weight, ids, pe_tx, by_gene)
            else:
                x_tx = F.gelu(torch.cat((emb(x_dict["tx"].long()).to(dt), pe_tx), -1))
        else:
            x_bd = F.gelu(x_bd)
            x_tx = F.gelu(emb(x_dict["tx"].long()).to(dt))
        x = {"tx": x_tx, "bd": x_bd}

        if graphs is None:       # sorted views of the edge stores: built once per batch, shared by all layers
            # the by-source view only serves the backward: inference sorts each edge store once, not twice
            # tx-belongs-bd: a transcript lies in at most one boundary (heterodata.py:147), so its backward needs no
            # by-source view; "lazy" verifies that on the device and sorts only if it does not hold (graph.EdgeGraph)
            by_src = {TX_TX: True, TX_BD: "lazy"} if torch.is_grad_enabled() else {TX_TX: False, TX_BD: False}
            graphs = {et: edge_graph(cache, et, edge_index_dict[et], x[et[0]].shape[0], x[et[2]].shape[0],
                                     need_by_src=by_src[et],
                                     validate="deferred")     # checked without a host sync (graph.py)
                      for et in (TX_TX, TX_BD) if et in edge_index_dict}
        step = self._step_dev
        seed_off = 0
        drawn = graphs.get("draws") if self.training else None
        if drawn is not None:
            # a captured step made all of its draws up front in one launch (ops.step_draws) from the counter as it stands
            # and advances it at its end: (planes | None, constant added to every seed)
            planes, seed_off = drawn
        elif self.training:
            # every training forward gets its own snapshot of the advanced counter: its backward re-reads THAT word,
            # so a second forward before the first backward (two views, checkpointing, a logging pass) cannot change
            # the masks the first backward regenerates.  Capture-safe: the clone lives in the graph's pool.
            if self._step_dev.is_cuda:
                step = ops.step_advance(self._step_dev, 256)
            else:
                self._step_dev.add_(256)
                step = self._step_dev.clone()
        if drawn is None:
            planes = self._dropout_planes(graphs, step) if self.training else None
        for li, layer in enumerate(self.conv_layers):
            kb = None if planes is None else {et: (d[li], None if s_ is None else s_[li]) for et, (d, s_) in planes.items()}
            x = layer(x, edge_index_dict, graphs=graphs, apply_gelu=True, seed=(li, step, seed_off), keep_bits=kb)   # conv + GELU (:324-325)

        x = self.lin_last(x)
        if self.normalize_embeddings:
            x = ops.l2_normalize_many(x)                     # both node types in one launch
        return x
