"""Functional stand-in for entmax (third-party, not installed here): SparsemaxLoss forwards to the oracle's restatement of the
published sparsemax loss, so a fixture with the 'segmentation' term pins the reference's WIRING (which prediction / ground truth
it reads, flattening, weight) — not entmax's arithmetic (PARITY UNPINNED at that call site, oracle/ref_path.py)."""
from oracle.ref_path import SparsemaxLoss  # noqa: F401
