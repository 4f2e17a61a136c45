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
             "readout_subgraph" in r["Name"]]
    steps = embed[0] if embed else 1
    print(f"{f}: total kernel time {tot/1e6:.2f} ms over {steps} steps = {tot/1e3/steps:.1f} us/step, "
          f"{sum(int(r['Calls']) for r in rows)/steps:.0f} kernels/step")
    for r in rows[:top]:
        print(f'{r["Name"][:86]:86s} n/step={int(r["Calls"])/steps:5.1f} us/step={float(r["TotalDurationNs"])/1e3/steps:9.1f} '
              f'avg_us={float(r["AverageNs"])/1e3:9.2f}')


if __name__ == "__main__":
    main()
