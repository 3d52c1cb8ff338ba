"""kspider_amd — MI355X-native pairwise containment engine behind kSpider's pairwise() surface.

Host-side mirror of the reference interface for this one path:
  kspider_amd.pairwise(index_prefix, user_threads)   == kSpider_internal.pairwise (kSpider_internal.i:11)
  kspider_amd.cluster(index_prefix, dist_type, cutoff)  == `kSpider cluster` (ks_clustering.py:150-163), components on the GPU
  kspider_amd.engine                                   ctypes binding of include/kspider_amd.h
  kspider_amd.dist                                     tile sharding + edge gather for one-process-per-GPU runs
  kspider_amd.synth                                    synthetic sketch sets shaped like BASELINE.json's configs
The compute lives in kspider_amd/lib/libkspider_amd.so (hand-written HIP, gfx950); nothing here
falls back to the CPU.
"""
from .engine import cluster, pairwise, pairwise_bins, pairwise_sigs  # noqa: F401

__all__ = ["pairwise", "pairwise_sigs", "pairwise_bins", "cluster"]
