#!/usr/bin/env python3
"""What the GPU was doing, from a rocprofv3 --kernel-trace CSV: per kernel name count / mean duration, time with no kernel running,
time with exactly one / two / three+ kernels running. usage: timeline.py <kernel_trace.csv> [t0_frac t1_frac]"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp")
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
t_lo, t_hi = df.Start_Timestamp.min(), df.End_Timestamp.max()
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
df = df[(df.Start_Timestamp >= a) & (df.End_Timestamp <= b)]
df["name"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 40)
df["dur"] = df.End_Timestamp - df.Start_Timestamp
print(df.groupby("name").dur.agg(["count", "mean", "sum"]).sort_values("sum", ascending=False).head(8).to_string())
ev = sorted([(t, 1) for t in df.Start_Timestamp] + [(t, -1) for t in df.End_Timestamp])
level, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[level] = hist.get(level, 0) + (t - last); last = t; level += d
span = ev[-1][0] - ev[0][0]
print(f"span {span / 1e6:.2f} ms; kernels running: " + ", ".join(f"{k}: {v / span * 100:.1f} %" for k, v in sorted(hist.items())))
tr = df[df.name.str.startswith("k_trace_pw")]
print(f"k_trace_pw: {len(tr)} launches, mean {tr.dur.mean() / 1e6:.3f} ms, median {tr.dur.median() / 1e6:.3f} ms")
for col in ("Queue_Id", "Stream_Id"):
    if col in df.columns:
        g = df[df.name.str.startswith(("k_trace_pw", "k_shade"))].groupby(col)
        print(f"by {col}: " + "; ".join(f"{k}: {len(v)} kernels, busy {v.dur.sum() / 1e6:.1f} ms, last end at {(v.End_Timestamp.max() - ev[0][0]) / span * 100:.1f} % of the span" for k, v in g))
