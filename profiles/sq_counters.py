"""Per-kernel SQ counter table from rocprofv3 --pmc passes.

    python profiles/sq_counters.py <name filter regex> <counter_collection.csv> [more csv ...]

Rows are (kernel name, grid size): the same template instance runs several layer shapes.  Values are
means per dispatch; ratios follow MI355X_MICROARCH.md (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, SQ_BUSY_CYCLES per SE).
"""
import csv
import re
import sys
from collections import defaultdict


def demangle(name):
    """rocprofv3 leaves names with bf16 template arguments (DF16b) mangled and this image has no demangler that knows
    them: rebuild `void dfd::kernel<args>` for this library's kernels (integer, bool, float and bf16 arguments)."""
    import re

    m = re.match(r"_ZN3dfd(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    ident = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    if not rest.startswith("I"):
        return "dfd::" + ident
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("Li", i) or rest.startswith("Lb", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append(("true" if v == "1" else "false") if rest[i + 1] == "b" else v.replace("n", "-"))
            i = j + 1
        elif rest.startswith("DF16b", i):
            args.append("__bf16")
            i += 5
        elif rest[i] == "f":
            args.append("float")
            i += 1
        else:
            return name
    return "void dfd::" + ident + "<" + ", ".join(args) + ">(...)"


def main():
    flt = re.compile(sys.argv[1])
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    meta = {}
    for path in sys.argv[2:]:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                name = demangle(r["Kernel_Name"])
                if not flt.search(name):
                    continue
                short = re.sub(r"^void |dfd::|\(.*$", "", name)
                key = (short, int(r["Grid_Size"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                meta[key] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    counters = sorted({c for v in acc.values() for c in v})
    print("| kernel | grid | vgpr | lds | us (profiled) | " + " | ".join(counters) + " |")
    print("|---|---|---|---|---|" + "---|" * len(counters))
    for key in sorted(acc):
        m = {c: sum(v) / len(v) for c, v in acc[key].items()}
        d = sorted(dur[key])[len(dur[key]) // 2]
        print(f"| `{key[0]}` | {key[1]} | {meta[key][0]} | {meta[key][1]} | {d:.1f} | "
              + " | ".join(f"{m.get(c, float('nan')):.4g}" for c in counters) + " |")


if __name__ == "__main__":
    main()
