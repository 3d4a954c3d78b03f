"""Instruction histogram of one kernel in a hipcc -S --cuda-device-only listing.

    python tools/isa_stats.py fused.s fused_iterate_kernelILi10E
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    hist = collections.Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        hist[t.split()[0]] += 1
    total = sum(hist.values())
    groups = collections.Counter()
    for k, v in hist.items():
        if k.startswith("v_") and "f64" in k:
            g = "valu_f64"
        elif k.startswith("v_"):
            g = "valu_other"
        elif k.startswith("ds_"):
            g = "lds"
        elif k.startswith("s_load") or k.startswith("s_buffer"):
            g = "smem"
        elif k.startswith("s_"):
            g = "salu"
        elif k.startswith(("global_", "flat_", "buffer_")):
            g = "vmem"
        elif k.startswith("scratch_"):
            g = "scratch"
        else:
            g = "other"
        groups[g] += v
    print(f"{key}: {total} static instructions")
    for g, v in groups.most_common():
        print(f"  {g:12s} {v}")
    for k, v in hist.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
        print(f"    {k:28s} {v}")
    for l in lines:
        if key in l and (".vgpr_count" in l or ".private_segment" in l):
            print(l)
    meta = "\n".join(lines)
    m = re.search(r"\.name:\s+_Z\w*" + re.escape(key) + r"\w*\n(?:.*\n){0,20}", meta)
    i = next((i for i, l in enumerate(lines) if ".name:" in l and key in l), None)
    if i is not None:
        for l in lines[max(0, i - 12):i + 14]:
            if any(s in l for s in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size",
                                    "group_segment_fixed_size", "spill")):
                print("   ", l.strip())


if __name__ == "__main__":
    main()
