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


def last_rows(path, n, substr=""):
  """The last `n` dispatches of the kernel with the largest total time: the timed
  region of a bench run (what comes before — Engine.place()'s dry launches of the
  same kernel, the reset, the warm-up — is left out)."""
  db = sqlite3.connect(path)
  cols = [c[1] for c in db.execute("pragma table_info(kernels)")] or \
         [d[0] for d in db.execute("select * from kernels limit 1").description]
  start = next((c for c in ("start", "start_timestamp", "begin", "start_ns") if c in cols), None)
  top = db.execute("select name from kernels where name like ? group by name "
                   "order by sum(duration) desc limit 1", (f"%{substr}%",)).fetchone()
  if start is None or top is None:
    return None
  d = [r[0] / 1e3 for r in db.execute(
      f"select duration from kernels where name = ? order by {start} desc limit ?", (top[0], n))]
  if not d:
    return None
  return short(top[0]), len(d), sum(d) / len(d), min(d), max(d)


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
  ap.add_argument("--bench-log", default="",
                  help="stdout of the traced bench.py run: its own JSON line is quoted next "
                       "to the trace (same process, same placement of the bound view)")
  ap.add_argument("--last-kernel", default="",
                  help="... of the kernel whose name contains this (default: the dominant one)")
  ap.add_argument("--bench-key", default="",
                  help="quote this sub-object of the bench line (e.g. substrate_api) instead of the line itself")
  ap.add_argument("--last", type=int, default=0,
                  help="also: the last N dispatches of the dominant kernel (the timed region)")
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
    if args.last:
      r = last_rows(args.trace, args.last, args.last_kernel)
      if r:
        lines += ["The last %d dispatches of `%s` (the timed region; before it: the dry "
                  "launches of `Engine.place()`, the reset, the warm-up): **avg %.2f µs**, "
                  "min %.2f, max %.2f." % (r[1], r[0], r[2], r[3], r[4]), ""]
  if args.bench_log:
    import json
    try:
      d = json.loads([l for l in open(args.bench_log).read().splitlines() if l.startswith("{")][-1])
      if args.bench_key:
        d = d[args.bench_key]
        lines += ["`bench.py`'s `%s` leg inside this traced process (HIP events around its %d timed "
                  "steps): **%.2f µs** per launch, %d launch(es) per step; %s" %
                  (args.bench_key, d["steps"], d["avg_launch_ms"] * 1e3, d["launches_per_step"],
                   {k: v for k, v in d.items() if k in ("value", "frac", "bytes_per_launch", "placement")}), ""]
      else:
        lines += ["`bench.py` inside this traced process (HIP events around its %d timed steps): "
                  "**%.2f µs** per launch; placement probe: %s." %
                  (d["steps"], d["kernels_ms"]["frame"] * 1e3, d.get("placement")), ""]
    except (OSError, IndexError, KeyError, ValueError):
      pass
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
