"""Repro driver for a flaky crash: run the body of test_graph_epochs_with_evaluation_between_match_eager_loop, then the eval
graph test, several times in one process with explicit garbage collections in between."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler

faulthandler.enable()
import pytest
import torch

import test_gpu_model as T


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("PROBE_KEEP", "0") == "1":  # never destroy an evaluation graph
    from glass_amd import evalstep
    _keep = []
    _orig = evalstep.EvalGraph.capture

    def _cap(self):
        r = _orig(self)
        _keep.append(self.graph)
        return r
    evalstep.EvalGraph.capture = _cap
for i in range(n):
    if os.environ.get("PROBE_EPOCHS", "1") == "1":
        T.test_graph_epochs_with_evaluation_between_match_eager_loop(MP())
        print("epochs test done", i, flush=True)
    if os.environ.get("PROBE_GC", "1") == "1":
        gc.collect()
        print("gc done", i, flush=True)
    T.test_eval_graph_parallel_branches_match_eager_forward()
    print("eval test done", i, flush=True)
print("ALL OK")
