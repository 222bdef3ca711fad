#!/usr/bin/env python3
"""Per-kernel averages of PMC counters from a rocprofv3 rocpd sqlite database."""
import re, sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
rows = cur.execute("select * from pmc_events limit 1").fetchall()
# generic: find name/counter/value columns
q = None
for cand in ("select k.name, p.counter_name, p.value from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id",
             "select name, counter_name, value from pmc_events",
             "select kernel_name, counter_name, counter_value from counters_collection"):
    try:
        res = cur.execute(cand).fetchall(); q = cand; break
    except Exception as e:
        pass
if q is None:
    print("pmc_events columns:", cols); print(rows); sys.exit(0)
agg = {}
for name, cn, v in res:
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    a = agg.setdefault((name, cn), [0, 0.0]); a[0] += 1; a[1] += float(v)
names = sorted({k[0] for k in agg}); cns = sorted({k[1] for k in agg})
print(f"{'kernel':72s} " + " ".join(f"{c[:22]:>22s}" for c in cns))
for n in names:
    print(f"{n:72s} " + " ".join(f"{agg.get((n,c),[1,0])[1]/max(1,agg.get((n,c),[1,0])[0]):22.0f}" for c in cns))
