"""ORACLE — test infrastructure only.  CPU restatement of the reference hot path (see ref_path.py) and
the C kNN definition (knn_ref.c).  Allowed importers: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product package never imports from here."""
