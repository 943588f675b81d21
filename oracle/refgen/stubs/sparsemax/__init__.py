"""Stand-in for the un-installed `sparsemax` package (see torch_geometric stub)."""
from oracle.ref_path import Sparsemax  # noqa: F401
