"""Drop-in for the reference's `lsh` extension module (library/lsh/lsh.cc:316-326)."""
from magicpig_amd.lsh import LSH  # noqa: F401
