"""CPU oracle for the hetero-GNN message-passing hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker (or the timed CPU baseline), never as the thing shipped.
The product path (``multi-modal-gnn_amd/``) raises if its HIP library is missing; it
never falls back to this code.

What it restates (all citations are into /root/reference/):
  * ``src/model.py:33-335``   HeteroRGCN (encode_nodes / forward / predict_lab_values)
  * ``src/model.py:342-396``  EdgeRegressionHead
  * ``src/model.py:579-612``  compute_regression_loss
  * ``src/graph_build.py:34-97,155-248,476-586``  NodeIndexer + edge_index builders
  * ``src/train.py:98-176,295-392``  edge splits, supervision mask, lab weights, weighted loss
  * ``src/evaluate.py:36-82,417-440``  regression metrics, per-lab winsorisation
  * torch_geometric (requirements.txt: ``torch-geometric>=2.3.0``, NOT vendored, NOT
    installed, no lock file): ``SAGEConv(aggr='mean')`` / ``HeteroConv(aggr='sum')`` /
    ``HeteroData`` restated from PyG's published semantics in ``pyg_min.py``.

Pinning status
  * Everything the reference itself owns (graph_build, EdgeMasker, lab weights,
    HeteroRGCN wiring, heads, loss) is pinned: ``oracle/gen_golden.py`` imports the
    reference's own modules in the build container (with ``pyg_min`` registered under
    the name ``torch_geometric``) and writes ``tests/golden/*.npz``; the oracle is
    checked against those fixtures in ``tests/test_oracle_golden.py``.
  * The arithmetic INSIDE SAGEConv/HeteroConv is third-party and absent, and the
    reference has no test that pins it: for that boundary the oracle is
    **parity unpinned** (restated from the published algorithm; anchored on the
    reference's call sites model.py:125-131,256 and on the published parameter counts
    483,970 / 465,409 / 415,873 which fix lin_l(bias)+lin_r(no bias) per relation).
"""
