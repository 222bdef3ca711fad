#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction and LDS bank-conflict fraction from two rocprofv3 --pmc passes (rocpd sqlite):
  pass 1: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      pass 2: --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
MFMA busy = sum over SIMDs of busy cycles / (1024 SIMDs * active cycles)."""
import re, sqlite3, sys
def load(db):
    con = sqlite3.connect(db); cur = con.cursor()
    rows = cur.execute("select name, counter_name, counter_value, duration, dispatch_id from pmc_events").fetchall()
    per = {}
    for n, c, v, dur, did in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)[:64]
        d = per.setdefault((n, did), {"dur": float(dur)})
        d[c] = d.get(c, 0.0) + float(v)
    agg = {}
    for (n, did), d in per.items():
        a = agg.setdefault(n, {"calls": 0, "dur": 0.0})
        a["calls"] += 1; a["dur"] += d["dur"]
        for k, v in d.items():
            if k != "dur": a[k] = a.get(k, 0.0) + v
    return agg
m, l = load(sys.argv[1]), load(sys.argv[2])
print(f"{'kernel':66s} {'calls':>6s} {'avg us':>9s} {'MFMA busy':>10s} {'LDS conflict/active':>20s}")
for n, a in sorted(m.items(), key=lambda kv: -kv[1]["dur"])[:24]:
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); act = a.get("GRBM_GUI_ACTIVE", 0.0)
    # GRBM_GUI_ACTIVE is reported once per XCD-SE group; normalise by the number of records per dispatch
    frac = busy / (1024.0 * act / max(1, round(act / max(1.0, a["dur"] * 2.1e-3 * a["calls"] / a["calls"])))) if act else 0.0
    lc = l.get(n, {})
    conf = lc.get("SQ_LDS_BANK_CONFLICT", 0.0); idx = lc.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{n:66s} {a['calls']:6d} {a['dur']/a['calls']/1e3:9.1f} {busy / (1024.0 * a['dur'] * 2.2) if a['dur'] else 0:10.3f} {conf/idx if idx else 0:20.4f}")
print("(MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (1024 SIMDs x duration x 2.2 GHz))")
