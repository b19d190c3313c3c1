#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch."""
import sys
import pandas as pd

frames = [pd.read_csv(p) for p in sys.argv[1:]]
df = pd.concat(frames)
df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
piv = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
pd.set_option("display.width", 250, "display.max_columns", 50, "display.float_format", lambda x: f"{x:,.0f}")
keep = piv.index.str.contains("qstream|qde|qfwd|qargmax|topk|adam|gather_pool|segsum")
print(piv[keep].to_string())
