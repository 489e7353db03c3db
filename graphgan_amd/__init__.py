"""graphgan_amd -- MI355X-native engine for the GraphGAN hot path (walk sampling, pair
reward, D/G update steps) behind the reference's ``graph_gan.py`` / ``config.py`` contract.

All numerics live in ``libgraphgan_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/graphgan_hip.h``); importing this package fails if the library is not built.
"""
from ._lib import (GG_EINVAL, GG_ECAPACITY, GG_EHIP, GG_ECOMM, GG_ENOMEM, GG_EIO,  # noqa: F401
                   GG_OPT_ADAM_DENSE, GG_OPT_ADAM_LAZY, GG_OPT_SGD, GG_ROOT_ABORTED, GG_ROOT_EMPTY, GG_ROOT_OK,
                   GraphGANHipError)
from .engine import (CSRGraph, Engine, edges_to_csr, read_edges_csr, graph_to_csr, host_build_trees, host_write_embeddings,  # noqa: F401
                     synth_powerlaw)
