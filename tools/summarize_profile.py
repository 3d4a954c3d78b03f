"""Turn the raw rocprofv3 outputs of tools/profile_gpu.sh (gpurun_out/<tag>/) into the
committed summaries profiles/<name>.md and profiles/<name>.json, and refresh
profiles/traffic.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import hashlib
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import code_object_meta

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
KERNEL = "fused_iterate_kernel"


def read_counters(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["_vgpr"] = [r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                                r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"]]
    return out


def kernel_source_hash():
    """Same digest as bench.py: the sources the timed kernel is built from."""
    h = hashlib.sha256()
    for f in ("mk_fused.hip", "mk_device.h"):
        with open(os.path.join(ROOT, "mink_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(tag, name, solves_per_launch, iterations=50):
    src = os.path.join(ROOT, "gpurun_out", tag)
    res = {"tag": tag, "kernel": KERNEL + "<10>", "solves_per_launch": solves_per_launch,
           "kernel_source_sha256": kernel_source_hash()}
    meta = code_object_meta.collect()      # from build/obj/*.o: the binary the run loaded was linked from these
    res["code_object"] = {k: v for k, v in meta.items() if k.startswith(KERNEL + "<10")}
    stats = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    lines = ["| kernel | calls | avg ns | total % |", "|---|---|---|---|"]
    for r in rows[:8]:
        lines.append(f"| {short(r['Name'])[:90]} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
        if KERNEL in r["Name"]:
            res["avg_ns"] = float(r["AverageNs"])
            res["calls"] = int(r["Calls"])
    c = {}
    for sub in ("pmc_inst", "pmc_act", "pmc_fetch", "pmc_write", "pmc_grbm"):
        for k, v in read_counters(os.path.join(src, sub)).items():
            c[k] = v if k == "_vgpr" else sum(v) / len(v)
    res["counters_per_launch"] = {k: v for k, v in c.items() if k != "_vgpr"}
    res["dispatch"] = dict(zip(["vgpr", "agpr", "sgpr", "lds_bytes", "scratch", "grid", "workgroup"], c.get("_vgpr", [])))
    n = solves_per_launch
    per = {}
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
              "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM"):
        if k in c:
            per[k] = c[k] * 64 / n   # wave instructions -> per-problem (lane) instruction count
    res["wave_instructions_per_solve"] = per
    if all(k in c for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")):
        flops = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"]) * 64
        res["executed_fp64_flop_per_solve"] = flops / n
        if "avg_ns" in res:
            res["executed_fp64_tflops"] = flops / res["avg_ns"] / 1e3
    # HBM traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
    # the bytes of wide coalesced reads (MI355X guide, HBM section) -> doubled here.
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd = 2.0 * c["FETCH_SIZE"] * 1024
        wr = c["WRITE_SIZE"] * 1024
        res["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr,
                                       "raw_fetch_kib": c["FETCH_SIZE"], "raw_write_kib": c["WRITE_SIZE"]}
        tj = {"hbm_bytes_per_launch": rd + wr, "problems_per_launch": n // iterations,
              "kernel_source_sha256": res["kernel_source_sha256"], "source": f"profiles/{name}.json",
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1`; "
                      "FETCH_SIZE doubled per the gfx950 correction; per launch of fused_iterate_kernel<10> over "
                      f"{n // iterations} problems x {iterations} iterations"}
        if "executed_fp64_flop_per_solve" in res:
            tj["executed_flop_per_solve"] = res["executed_fp64_flop_per_solve"]
        with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
            json.dump(tj, f, indent=1)
    # second trace: every workload of the bench, each kernel with what its code object carries
    allk = []
    stats_all = glob.glob(os.path.join(src, "kt_all", "**", "*_kernel_stats.csv"), recursive=True)
    if stats_all:
        for r in list(csv.DictReader(open(stats_all[0])))[:14]:
            nm = short(r["Name"])
            m = meta.get(nm, {})
            allk.append({"kernel": nm, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                         "percent": float(r["Percentage"]), "vgpr": m.get("vgpr_count"), "agpr": m.get("agpr_count"),
                         "vgpr_spills": m.get("vgpr_spill_count"), "scratch_bytes_per_lane": m.get("private_segment_fixed_size"),
                         "static_lds": m.get("group_segment_fixed_size")})
        res["all_workloads_kernels"] = allk
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", name + ".json"), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(ROOT, "profiles", name + ".md"), "w") as f:
        f.write(f"# {name}: rocprofv3 summary of `python bench.py --steps 3 --warmup 1` (1 x MI355X)\n\n")
        f.write("## --kernel-trace --stats\n\n" + "\n".join(lines) + "\n\n")
        f.write("## dispatch (as recorded by the profiler)\n\n" + json.dumps(res["dispatch"]) + "\n\n")
        f.write("## code object (build/obj, llvm-readelf --notes)\n\n"
                + code_object_meta.table(res["code_object"]) + "\n\n"
                "The profiler's dispatch line shows the architectural VGPR half only (granule-rounded) and no "
                "dynamic LDS; the code object's vgpr_count is the unified total (architectural + AGPR), which is "
                "what limits the kernel to one wave per SIMD. Dynamic LDS of the launch: 64 x ((nq | 1) + "
                "NC (NC + 1) / 2) x 8 B = 44032 B per block for TidyBot (nq 31, NC 10); four blocks are resident "
                "per CU (adding 4 KB per block drops the kernel to three and its time by 4/3, measured).\n\n")
        f.write("## PMC (separate passes), mean per launch of the iterate kernel\n\n")
        for k, v in sorted(res["counters_per_launch"].items()):
            f.write(f"- {k}: {v:.6g}\n")
        f.write("\n## derived\n\n")
        for k in ("wave_instructions_per_solve", "executed_fp64_flop_per_solve", "executed_fp64_tflops",
                  "hbm_bytes_per_launch"):
            if k in res:
                f.write(f"- {k}: {json.dumps(res[k])}\n")
        if allk:
            f.write("\n## all bench workloads (`bench.py --steps 3 --warmup 1 --no-cpu-baseline`), --kernel-trace --stats\n\n")
            f.write("| kernel | calls | avg ns | total % | vgpr | agpr | vgpr spills | scratch B/lane | static LDS B |\n|---|---|---|---|---|---|---|---|---|\n")
            for k in allk:
                f.write(f"| {k['kernel'][:80]} | {k['calls']} | {k['avg_ns']:.0f} | {k['percent']} | {k['vgpr']} | {k['agpr']} | "
                        f"{k['vgpr_spills']} | {k['scratch_bytes_per_lane']} | {k['static_lds']} |\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 20) * 50)
