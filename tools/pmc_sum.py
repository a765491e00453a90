#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: pmc_sum.py <dir or csv> [kernel substring]"""
import glob
import os
import sys

import pandas as pd

src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
df = pd.concat([pd.read_csv(f) for f in files])
if sub:
    df = df[df.Kernel_Name.str.contains(sub, regex=False)]
df["k"] = df.Kernel_Name.str.slice(0, 40)
t = df.groupby(["k", "Counter_Name"]).Counter_Value.agg(["sum", "count"])
pd.set_option("display.width", 200)
print(t.to_string())
