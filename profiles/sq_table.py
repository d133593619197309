"""Derived per-kernel table from the two SQ PMC passes of profiles/run_profiles.sh:

    python profiles/sq_table.py <pmc_sqa csv> <pmc_sqb csv> > profiles/rNN_sq_counters.md

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the 1024 SIMDs; SQ_LDS_IDX_ACTIVE / _BANK_CONFLICT cycles summed over the
256 CUs.  "busy" columns divide by (kernel duration x 2.4 GHz x units): an upper-clock estimate (the chip runs below
2.4 GHz under load, so real utilisation is somewhat higher)."""
import csv
import re
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from sq_counters import demangle  # noqa: E402

acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = demangle(r["Kernel_Name"])
        if "dfd::" not in name:
            continue
        key = (re.sub(r"^void |dfd::|\(.*$", "", name), int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | grid | us | VALU busy | MFMA busy | LDS busy | LDS conflict share | active / wave-cyc | issue-stall / wave-cyc | wait / wave-cyc | VALU insts / wave |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
rows = []
for key, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    us = sorted(dur[key])[len(dur[key]) // 2]
    cyc = us * 2400.0
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    waves = m.get("SQ_WAVES", 0) or 1
    rows.append((us, f"| `{key[0]}` | {key[1]} | {us:.1f} | {m.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (1024 * cyc):.2f} | "
                     f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.2f} | {m.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc):.2f} | "
                     f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 0), 1):.2f} | "
                     f"{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {m.get('SQ_WAIT_ANY', 0) / wc:.2f} | "
                     f"{m.get('SQ_INSTS_VALU', 0) / waves:.0f} |"))
for _, line in sorted(rows, key=lambda r: -r[0]):
    print(line)
