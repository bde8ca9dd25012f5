#!/usr/bin/env python3
"""Extract the curve tables the reference's notebook stores as cell outputs (notebooks/intro.ipynb cells 12 and 20:
`OISCurve.__repr__` of the README curve and of its 10Y + 1 bp scenario) into tests/golden/notebook_curve_tables.json.
The notebook is data the reference ships with its own outputs; run this where /root/reference exists."""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/notebooks/intro.ipynb"
nb = json.load(open(src))
out = {}
for name, cell in (("base", 12), ("bump_10Y_1bp", 20)):
    text = "".join("".join(o["text"]) for o in nb["cells"][cell]["outputs"] if "text" in o and "CURVE DETAILS" in "".join(o["text"]))
    rows = []
    for line in text.splitlines():
        cols = [c.strip() for c in line.strip().strip("|").split("|")]
        try:
            rows.append([float(c) for c in cols])
        except ValueError:
            continue
    out[name] = {"columns": ["tenor_years", "last_year_fraction", "rate", "df"], "rows": rows}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "notebook_curve_tables.json")
json.dump(out, open(path, "w"), indent=1)
print({k: len(v["rows"]) for k, v in out.items()})
