"""hsearch_amd -- MI355X-native protein-motif LSH search (the hot path of acgtun/hsearch).

The product is the C-ABI shared library ``libhsearch_amd.so`` (include/hsearch.h) built from the
hand-written gfx950 kernels in ``hsearch_amd/csrc``.  This package is a thin ctypes binding used by
the tests and bench.py; it contains no compute and no CPU fallback: if the library is missing or
no gfx950 device is present, calls raise.
"""
from .capi import (KLSH_NONE, ClusterState, Engine, HsError, alphabet, clustering, clusters_file_text,
                   codes_from_letters, index_file_check, key_string, klsh_codes, klsh_draw_planes, lib_path, load,
                   profile_fields)

__all__ = ["KLSH_NONE", "ClusterState", "klsh_codes", "klsh_draw_planes", "Engine", "HsError", "alphabet", "clustering", "clusters_file_text", "codes_from_letters", "index_file_check", "key_string", "lib_path", "load",
           "profile_fields"]
