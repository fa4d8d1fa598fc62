"""MI355X-native heterogeneous-GNN message passing: drop-in for the hot path of
AdalineL/Multi-Modal-GNN (``src/model.py`` HeteroRGCN on the ``edge_index`` contract of
``src/graph_build.py``).  Python host over hand-written HIP kernels (libmmgnn.so, C ABI in
``include/mmgnn.h``).  There is no CPU fallback: the ops raise if the library is missing.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # lazy: keep `import mmgnn` cheap and free of torch for tools that only need the C ABI table
    if name in ("build_model", "HeteroRGCN", "EdgeRegressionHead", "compute_regression_loss"):
        from . import model
        return getattr(model, name)
    if name in ("HeteroGraph",):
        from . import data
        return getattr(data, name)
    raise AttributeError(name)
