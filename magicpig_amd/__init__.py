"""magicpig_amd -- MI355X-native (gfx950) LSH-sampled sparse decode attention.

The product is the C-ABI shared library declared in include/magicpig_hip.h
(magicpig_amd/lib/libmagicpig_hip.so, hand-written HIP); this package is the host-side mirror
of the reference's operator API over it:

    from magicpig_amd import LSH, SparseAttentionServer        # library/lsh, library/sparse_attention
    from magicpig_amd import SimHash, LSHSparseAttnServer      # models/attnserver.py hot path

Drop-in module names `lsh` and `sparse_attention_cpu` live in magicpig_amd/dropin (put that
directory on PYTHONPATH, or call magicpig_amd.install_dropin()).
"""
from .lsh import LSH  # noqa: F401
from .simhash import SimHash  # noqa: F401
from .sparse_attention import SparseAttentionServer  # noqa: F401
from .attnserver import LSHSparseAttnServer  # noqa: F401
from ._lib import MagicPigError  # noqa: F401


def install_dropin() -> None:
    """Register modules named `lsh` and `sparse_attention_cpu` (models/attnserver.py:3-4)."""
    import sys

    from .dropin import lsh as _l, sparse_attention_cpu as _s

    sys.modules.setdefault("lsh", _l)
    sys.modules.setdefault("sparse_attention_cpu", _s)
