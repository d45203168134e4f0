#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into a small markdown file.

  python tools/rocprof_summary.py --trace gpurun_out/prof_trace/r1_results.db \
      [--pmc gpurun_out/prof_fetch/r1_results.db ...] --out profiles/r01_x.md \
      [--note "..."]

--trace: a `rocprofv3 --kernel-trace --stats` database -> per-kernel calls,
         total / average / min / max duration.
--pmc:   `rocprofv3 --pmc <COUNTER>` databases (one counter set per pass, as
         MI355X_MICROARCH.md prescribes) -> per-kernel mean counter value.
"""
import argparse
import sqlite3


def short(name: str) -> str:
  name = name.replace("(anonymous namespace)::", "")
  if len(name) > 90:
    name = name[:87] + "..."
  return name


def trace_rows(path):
  db = sqlite3.connect(path)
  q = ("select name, count(*), sum(duration), avg(duration), min(duration), "
       "max(duration) from kernels group by name order by sum(duration) desc")
  rows = list(db.execute(q))
  total = sum(r[2] for r in rows) or 1
  return [(short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total)
          for n, c, s, a, mn, mx in rows]


def pmc_rows(path):
  db = sqlite3.connect(path)
  q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) "
       "from counters_collection group by kernel_name, counter_name "
       "order by sum(value) desc")
  return [(short(n), c, k, a, s) for n, c, k, a, s in db.execute(q)]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--trace")
  ap.add_argument("--pmc", nargs="*", default=[])
  ap.add_argument("--out", required=True)
  ap.add_argument("--title", default="rocprofv3 summary")
  ap.add_argument("--note", action="append", default=[])
  args = ap.parse_args()
  lines = [f"# {args.title}", ""]
  for n in args.note:
    lines += [n, ""]
  if args.trace:
    lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats`)", "",
              "| kernel | calls | total µs | avg µs | min µs | max µs | % |",
              "|---|---:|---:|---:|---:|---:|---:|"]
    for r in trace_rows(args.trace)[:12]:
      lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % r)
    lines.append("")
  for p in args.pmc:
    lines += [f"## counters (`rocprofv3 --pmc`, {p.split('/')[-2]})", "",
              "| kernel | counter | dispatches | mean per dispatch | sum |",
              "|---|---|---:|---:|---:|"]
    for r in pmc_rows(p)[:10]:
      lines.append("| `%s` | %s | %d | %.1f | %.1f |" % r)
    lines.append("")
  with open(args.out, "w") as f:
    f.write("\n".join(lines))
  print("\n".join(lines))


if __name__ == "__main__":
  main()
