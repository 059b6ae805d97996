"""million_amd — MI355X-native (gfx950) PQ-KV attention hot path of MILLION.

Layout: csrc/ (hand-written HIP kernels + C ABI, built into libmillion_hip.so by `make bindings`),
ops.py (torch tensors -> C ABI), pq_cache.py (host mirror of the reference's DynamicPQCache /
PagedPQCache / PageManager / KernelRegistry).  Importing the package does not load the library;
the first op does, and raises if it is missing (no CPU fallback)."""

__all__ = ["ops", "pq_cache", "build"]
