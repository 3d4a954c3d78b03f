"""Register / spill / scratch figures of the gfx950 kernels, read from the code objects the
build produced (build/obj/*.o): what the binary carries, as opposed to what a profiler's
dispatch record shows.

    python tools/code_object_meta.py [substring ...]       # table on stdout
    python tools/code_object_meta.py --json out.json

Used by tools/summarize_profile.py to put the figures into the profile summaries.
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n")[:len(names)]


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.elf")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj],
                              stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    res = []
    for block in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
        block = ".agpr_count:" + block
        rec = {}
        for f in FIELDS + ("name",):
            m = re.search(r"\." + f + r":\s+(\S+)", block)
            if m:
                rec[f] = m.group(1) if f == "name" else int(m.group(1))
        if "name" in rec:
            res.append(rec)
    for rec, dm in zip(res, demangle([r["name"] for r in res])):
        rec["demangled"] = re.sub(r"\(anonymous namespace\)::", "", dm).split("(")[0].replace("void ", "")
        if re.search(r"\breal\b", dm):      # the float32 build of the fused kernels takes `real` scalars
            rec["demangled"] += " [float32]"
    return res


def collect():
    out = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.o"))):
        try:
            for k in kernels_of(obj):
                out[k["demangled"]] = {f: k.get(f) for f in FIELDS}
        except subprocess.CalledProcessError:
            pass
    return out


def table(meta, filters=()):
    rows = ["| kernel | vgpr | agpr | sgpr | vgpr spills | sgpr spills | scratch B/lane | static LDS B |",
            "|---|---|---|---|---|---|---|---|"]
    for name, m in meta.items():
        if filters and not any(f in name for f in filters):
            continue
        rows.append(f"| {name} | {m['vgpr_count']} | {m['agpr_count']} | {m['sgpr_count']} | {m['vgpr_spill_count']} | "
                    f"{m['sgpr_spill_count']} | {m['private_segment_fixed_size']} | {m['group_segment_fixed_size']} |")
    return "\n".join(rows)


if __name__ == "__main__":
    meta = collect()
    if len(sys.argv) > 2 and sys.argv[1] == "--json":
        json.dump(meta, open(sys.argv[2], "w"), indent=1)
    else:
        print(table(meta, sys.argv[1:]))
