"""protocol_amd — MI355X-native job-to-worker matching engine for the allocation hot path of the
PrimeIntellect-ai/protocol orchestrator.

The product is libpm_engine.so (HIP kernels + C ABI, include/pm_engine.h).  This package only holds
the pieces the path needs around it: the in-tree build, the ctypes binding, host-side packing and the
synthetic swarm generator used by tests and bench.py.
"""
__all__ = ["build", "engine", "host", "swarm"]
