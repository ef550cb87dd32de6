"""segger_amd: MI355X-native hot path of dpeerlab/segger.

Heterogeneous transcript<->boundary GATv2 message passing and the
transcript->cell edge-scoring heads, as hand-written gfx950 HIP kernels behind a
C ABI (``include/segger_amd.h`` -> ``segger_amd/libsegger_amd.so``), with
host-side mirrors of the reference's ``ISTEncoder`` and ``LitISTEncoder``.
"""
from .hetero import HeteroBatch, collate, TX_TX, TX_BD, TX_NB_BD  # noqa: F401
from .ist_encoder import ISTEncoder, SkipGAT, Positional2dEmbedder, GATv2Conv  # noqa: F401
from .lightning_model import LitISTEncoder  # noqa: F401

__version__ = "0.1.0"
