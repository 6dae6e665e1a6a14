#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (short name) mean counter values per dispatch."""
import csv, sys, glob, collections, re
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    rows = list(csv.DictReader(open(f[0])))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"]
        m = re.search(r"(conv3x3_v3_kernel<[^>]*>|conv3x3_kernel<[^>]*>|stem_kernel<\d>|decoder_kernel<\d>|median_kernel|\w+_kernel)", name)
        acc[m.group(1) if m else name[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "conv3x3" not in k and "stem" not in k and "decoder" not in k: continue
        print(k, "dispatches", len(next(iter(cs.values()))))
        for c, v in cs.items():
            print(f"    {c:28s} mean {sum(v)/len(v):16.1f}")
