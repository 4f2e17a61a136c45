"""Per-step kernel breakdown from a rocprofv3 `--kernel-trace --stats --output-format csv` run of bench.py.
usage: python tools/prof_summary.py <dir containing *_kernel_stats.csv> [top_n]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    f = glob.glob(f"{d}/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    embed = [int(r["Calls"]) for r in rows if "embed_label" in r["Name"] or "embed_gather" in r["Name"] or
             "readout_subgraph" in r["Name"] or "pair_head_fwd" in r["Name"]]
    steps = embed[0] if embed else 1
    # the step's own kernels: those launched at least once per step (setup kernels and bench.py's spin kernel — which only
    # queues work behind a sleep so that timed replays are not launch-bound — are listed but not counted)
    own = [r for r in rows if int(r["Calls"]) >= steps]
    own_t = sum(float(r["TotalDurationNs"]) for r in own)
    print(f"{f}: {steps} steps; the step's kernels: {own_t/1e3/steps:.1f} us/step in {sum(int(r['Calls']) for r in own)/steps:.1f} "
          f"launches/step (all kernels of the process: {tot/1e6:.2f} ms)")
    for r in rows[:top]:
        print(f'{r["Name"][:86]:86s} n/step={int(r["Calls"])/steps:5.1f} us/step={float(r["TotalDurationNs"])/1e3/steps:9.1f} '
              f'avg_us={float(r["AverageNs"])/1e3:9.2f}')


if __name__ == "__main__":
    main()
