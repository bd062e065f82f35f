"""Median kernel durations (us) from a rocprofv3 results .db: python tools/kstat.py file.db [name-substring ...]"""
import re, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
by = {}
for n, s, e in rows:
    by.setdefault(re.sub(r'\(.*', '', n).replace('void gv::', '').replace('gv::', ''), []).append((e - s) / 1e3)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(p in n for p in sys.argv[2:]):
        continue
    print(f"{n[:70]:70s} n={len(d):5d} median {statistics.median(d):8.1f} min {min(d):8.1f} max {max(d):8.1f} total {sum(d) / 1e3:8.2f} ms")
