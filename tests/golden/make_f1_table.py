"""Task-level reference numbers: RUN THE REFERENCE'S OWN DRIVER (/root/reference/GLASSTest.py:178-269, unmodified,
CPU, `--device -1`) for `--repeat R` repeats of a shipped synthetic set and record what it prints per repeat
(`end: epoch E, train time T s, val V, tst S`, GLASSTest.py:262-265) into tests/golden/g12_f1_<dataset>_<feature>.npz.

Authoring container only (needs /root/reference); PyG symbols come from pyg_stub.py exactly as for make_golden.py.
The fixture holds numbers the reference printed — per-repeat epochs / validation / test micro-F1 — and the command
line; no reference source.  The GPU-side counterpart is tools/f1_table.py, the test tests/test_gpu_task_parity.py.

    python tests/golden/make_f1_table.py --dataset density --feature use_one --repeat 10 [--threads 1]
"""
import argparse
import io
import os
import re
import runpy
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class Tee(io.TextIOBase):
    def __init__(self, real):
        self.real, self.lines, self._buf = real, [], ""

    def write(self, s):
        self.real.write(s)
        self._buf += s
        while "\n" in self._buf:
            line, self._buf = self._buf.split("\n", 1)
            self.lines.append(line)
        return len(s)

    def flush(self):
        self.real.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", required=True, choices=["density", "cut_ratio", "coreness", "component"])
    ap.add_argument("--feature", default="use_one", choices=["use_one", "use_deg"])
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()

    sys.dont_write_bytecode = True
    sys.path.insert(0, HERE)
    sys.path.insert(0, REF)
    import pyg_stub
    pyg_stub.install()
    os.chdir(REF)
    import torch
    torch.set_num_threads(a.threads)

    argv = ["GLASSTest.py", f"--{a.feature}", "--use_seed", "--use_maxzeroone", "--repeat", str(a.repeat), "--device", "-1",
            "--dataset", a.dataset]
    sys.argv = argv
    tee = Tee(sys.stdout)
    sys.stdout = tee
    t0 = time.time()
    try:
        runpy.run_path(os.path.join(REF, "GLASSTest.py"), run_name="__main__")
    finally:
        sys.stdout = tee.real
    wall = time.time() - t0
    end = re.compile(r"^end: epoch (\d+), train time ([0-9.]+) s, val ([0-9.]+), tst ([0-9.]+)")
    rows = [end.match(ln).groups() for ln in tee.lines if end.match(ln)]
    assert len(rows) == a.repeat, (len(rows), a.repeat)
    avg = [ln for ln in tee.lines if ln.startswith("average ")]
    out = os.path.join(HERE, f"g12_f1_{a.dataset}_{a.feature}.npz")
    np.savez_compressed(out, epochs=np.array([int(r[0]) for r in rows]), train_seconds=np.array([float(r[1]) for r in rows]),
                        val=np.array([float(r[2]) for r in rows]), tst=np.array([float(r[3]) for r in rows]),
                        command=np.array(" ".join(argv)), summary=np.array(avg[-1] if avg else ""),
                        threads=np.array(a.threads), wall_seconds=np.array(wall), torch_version=np.array(torch.__version__))
    print(f"wrote {out}: tst {[float(r[3]) for r in rows]}  ({wall:.0f} s)")


if __name__ == "__main__":
    main()
