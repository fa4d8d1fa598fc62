"""bench.py against another build of the library (A/B on one box): MMG_AB_LIB=ab/libmmgnn_<x>.so python profiles/probes/bench_ab.py <bench args>"""
import os, sys, runpy
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import mmgnn  # noqa: F401
from mmgnn import _lib
if os.environ.get("MMG_AB_LIB"):
    _lib.LIB_PATH = os.path.join(REPO, os.environ["MMG_AB_LIB"])
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(REPO, "bench.py"), run_name="__main__")
