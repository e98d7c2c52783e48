"""Drop-in for the reference's `sparse_attention_cpu` extension module
(library/sparse_attention/sparse_attention.cc:1243-1263); the state lives in HBM."""
from magicpig_amd.sparse_attention import SparseAttentionServer  # noqa: F401
