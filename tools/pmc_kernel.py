"""Sums rocprofv3 --pmc counters per kernel name from one or more result databases: python tools/pmc_kernel.py <match> <db> [...]"""
import sqlite3, sys
from collections import defaultdict
match = sys.argv[1]
for db in sys.argv[2:]:
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        print(db, "no counters_collection; tables:", tabs[:20]); continue
    acc, cnt = defaultdict(float), defaultdict(set)
    for did, kn, cn, v in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        if match in kn:
            key = kn.split("(")[0][-40:]
            acc[(key, cn)] += v
            cnt[(key, cn)].add(did)
    for (k, cn), v in sorted(acc.items()):
        print(f"{k:42s} {cn:28s} {v / len(cnt[(k, cn)]):16.1f}  per dispatch ({len(cnt[(k, cn)])} dispatches)")
