"""Inert stand-in for libigl bindings (data pipeline only)."""
