#!/usr/bin/env python3
"""Per-kernel averages of the hardware counters in rocprofv3 --pmc result databases (rocpd sqlite).

    python tools/pmc_summary.py <dir or .db> [...] [--match SUBSTR]

Prints one JSON object {kernel: {counter: mean value per dispatch, "dispatches": n}}.  FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts half of the bytes read (MI355X_MICROARCH.md, HBM section) -- the summary adds
"hbm_read_bytes" = FETCH_SIZE * 1024 * 2 and "hbm_write_bytes" = WRITE_SIZE * 1024 where those counters are present.
"""
import glob
import json
import os
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    dbs = []
    for a in args:
        dbs += [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
    acc = {}
    for path in dbs:
        db = sqlite3.connect(path)
        q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id"
        for name, counter, _, value in db.execute(q):
            if match and match not in name:
                continue
            short = name.split("(")[0].strip()
            acc.setdefault(short, {}).setdefault(counter, []).append(float(value))
    out = {}
    for k, cs in acc.items():
        row = {c: sum(v) / len(v) for c, v in cs.items()}
        row["dispatches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in row:
            row["hbm_read_bytes"] = row["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in row:
            row["hbm_write_bytes"] = row["WRITE_SIZE"] * 1024
        out[k] = row
    for a in sys.argv[1:]:                      # --set key=value: header fields of the summary (round, batch, command)
        if a.startswith("--set="):
            k, v = a[6:].split("=", 1)
            out[k] = int(v) if v.isdigit() else v
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
