"""CPU oracle for the MagicPIG hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the timed CPU baseline.  Nothing under
``magicpig_amd/`` imports it.
"""
from .oracle import (  # noqa: F401
    LSH,
    SparseAttentionServer,
    build,
    centre_keys,
    lib,
    merge_state,
    simhash_keys,
    simhash_query,
)
