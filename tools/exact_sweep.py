"""Step time of the step program with / without the exact GraphNorm accumulators over graph size and width:
    GLASS_GN_EXACT=0|1 python tools/exact_sweep.py <n_node> <n_pairs> <hidden> <layers>   (tools/exact_sweep.sh runs the sweep)
The em_user workload entry is patched to the requested shape and bench.py's own timing is used."""
import dataclasses, json, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glass_amd import synth
n, pairs, hid, layers = (int(a) for a in sys.argv[1:5])
synth.WORKLOADS["em_user"] = dataclasses.replace(synth.WORKLOADS["em_user"], n_node=n, n_pairs=pairs, hidden=hid, layers=layers)
import bench
sys.argv = ["bench.py", "--workload", "em_user", "--steps", "100", "--warmup", "10", "--min-blocks", "9", "--no-cpu-baseline", "--no-roofline-hbm", "--no-pmc"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"N={n} pairs={pairs} H={hid} L={layers} GN_EXACT={os.environ.get('GLASS_GN_EXACT','1')}: {d['ms_per_step']:.4f} ms")
