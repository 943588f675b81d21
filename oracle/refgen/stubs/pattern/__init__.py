"""Inert stand-in for Garment-Pattern-Generator's `pattern` package (data pipeline only)."""
from . import core, wrappers, rotation  # noqa: F401
