"""Stand-in for the un-installed torch_geometric package, used ONLY by oracle/refgen/make_golden.py
to make /root/reference/nn importable in the build container.  The arithmetic lives in
oracle/ref_path.py (restated from the published definitions; parity unpinned vs real PyG)."""
from . import nn  # noqa: F401
