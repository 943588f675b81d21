"""Inert stand-in for Garment-Pattern-Generator's customconfig (data pipeline only)."""


class Properties:
    def __init__(self, *a, **kw):
        raise NotImplementedError
