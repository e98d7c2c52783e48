"""Modules with the reference's import names (`from lsh import LSH`,
`from sparse_attention_cpu import SparseAttentionServer`, models/attnserver.py:3-4)."""
