"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --steps K --warmup W --no-cpu-baseline --no-e2e`) into profiles/traffic.json, the
`roofline.traffic` bench.py reports, and a per-kernel table.

    python profiles/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps+warmup> <tag>

gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE reports exactly
half the bytes of a wide coalesced (16 B/lane) stream, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane stores.  The depthwise kernels with 16-channel chunks (CB=16) read 64-byte pieces,
a width the guide calls uncalibrated - their doubled figure is an upper bound."""
import collections
import csv
import json
import os
import sys


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


def load(path, name):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            a = agg[demangle(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


def main():
    fetch_csv, write_csv, forwards, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    out_json = sys.argv[5] if len(sys.argv) > 5 else "traffic.json"
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows, dw_bytes = [], 0.0
    for k in sorted(f, key=lambda k: -f[k][0]):
        if not k.startswith(("void dfd::", "dfd::")):
            continue
        fe = 2.0 * f[k][0] * 1024 / forwards                 # bytes per forward, corrected
        wr = w.get(k, [0.0, 1])[0] * 1024 / forwards
        rows.append({"kernel": k.split("(")[0], "launches_per_forward": f[k][1] / forwards,
                     "fetch_bytes_per_forward": fe, "write_bytes_per_forward": wr})
        if "dw_kernel" in k or "mbconv" in k or "dw_rows7" in k:             # fused stem/expand + depthwise launches are depthwise launches
            dw_bytes += fe + wr
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import source_stamp

    json.dump({"stamp": source_stamp.stamp(),
               "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {forwards} forwards of batch 256 ({tag})",
               "correction": "FETCH_SIZE x2 (gfx950), KiB -> bytes", "dw_hbm_bytes_per_step": round(dw_bytes),
               "per_kernel": rows}, open(os.path.join(here, out_json), "w"), indent=1)
    with open(os.path.join(here, f"{tag}_pmc_traffic.md"), "w") as o:
        o.write(f"# HBM traffic per forward (batch 256), {tag}\n\n| kernel | launches | fetch MB (x2 corrected) | write MB |\n|---|---|---|---|\n")
        for r in rows:
            o.write(f"| `{r['kernel']}` | {r['launches_per_forward']:.0f} | {r['fetch_bytes_per_forward'] / 1e6:.1f} | {r['write_bytes_per_forward'] / 1e6:.1f} |\n")
        o.write(f"\ndepthwise family total: {dw_bytes / 1e9:.3f} GB per forward (algorithmic 6.427 GB)\n")
    print("dw_hbm_bytes_per_step", round(dw_bytes))


if __name__ == "__main__":
    main()
