"""gpurun_out/prof_<tag>/ -> profiles/<round>/<tag>_* and profiles/traffic.json (per-kernel HBM bytes per launch).

usage: python tools/collect_profiles.py <tag> <round dir, e.g. r01>
FETCH_SIZE / WRITE_SIZE are reported in KiB units per dispatch (MI355X_MICROARCH.md, rocprofv3 section).  FETCH_SIZE is
DOUBLED: tools/probes/fetch_calib.hip (profiles/r02/fetch_calibration.txt) shows the gfx950 counter reports one half of
the bytes for every access width these kernels use, including k_sample's row-segment footprint pattern; WRITE_SIZE is
exact (also cross-checked on k_pack_frame, whose output is exactly rows*cols*4 bytes).  When a pmc_sq pass exists the
per-launch VALU instruction count goes into traffic.json too (bench.py turns it into a VALU-issue floor).
"""
import collections, csv, glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return {k: 1024.0 * v[0] / v[1] for k, v in acc.items()}


fetch_csv, write_csv = one("pmc_fetch/**/*counter_collection.csv"), one("pmc_write/**/*counter_collection.csv")
fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")


def slim(path, out):   # keep the pba:: kernels only (the raw files are several MB)
    rows = [r for r in csv.DictReader(open(path)) if "pba::" in r["Kernel_Name"]]
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                          "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
                                          "Counter_Value"], extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)


slim(fetch_csv, os.path.join(dst, tag + "_pmc_fetch_size.csv"))
slim(write_csv, os.path.join(dst, tag + "_pmc_write_size.csv"))
valu = {}
sq = glob.glob(os.path.join(src, "pmc_sq/**/*counter_collection.csv"), recursive=True)
if sq:
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(sq[0])):
        if "pba::" not in r["Kernel_Name"]:
            continue
        a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(dst, tag + "_pmc_sq.txt"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                "SQ_INSTS_LDS SQ_INSTS_VMEM_RD -- python bench.py --no-cpu-baseline (per-launch averages)\n")
        for k, cs in acc.items():
            n = max(v[1] for v in cs.values())
            if n < 5:
                continue
            f.write("%s (%d launches)\n" % (k, n))
            for c, v in sorted(cs.items()):
                f.write("    %-26s %14.0f\n" % (c, v[0] / v[1]))
            if "SQ_INSTS_VALU" in cs:
                valu[k] = cs["SQ_INSTS_VALU"][0] / cs["SQ_INSTS_VALU"][1]
per = {k: {"fetch_bytes": 2.0 * fetch.get(k, 0.0), "write_bytes": write.get(k, 0.0),
           "total_bytes": 2.0 * fetch.get(k, 0.0) + write.get(k, 0.0), "valu_insts": valu.get(k)}
       for k in sorted(set(fetch) | set(write)) if "pba::" in k}
sys.path.insert(0, os.getcwd())
import bench  # noqa: E402  (kernel_source_id: ties the counters to the build they were taken on)
out = {
    "kernel_source_id": bench.kernel_source_id(),
    "unit": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 from separate rocprofv3 --pmc passes; the x2 on "
            "FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md, calibrated on these kernels' own access widths "
            "(profiles/r02/fetch_calibration.txt: ratio 0.500 for dword..dwordx4 streams, 0.532 for the footprint row "
            "segments); WRITE_SIZE is exact; valu_insts = SQ_INSTS_VALU per launch (wave instructions)",
    "workload": "configs[1]: 8 frames x 50k points, R=2, 400k observations",
    "source": "%s/%s_pmc_*.csv" % (dst, tag),
    "per_kernel": per,
}
for k, v in per.items():
    if "k_sample<2, true, 4, true, true, false>" in k:
        out["k_sample<JAC>"] = v["total_bytes"]
        out["k_sample<JAC>_valu_insts"] = v["valu_insts"]
    if k.endswith("k_schur"):
        out["k_schur"] = v["total_bytes"]
        out["k_schur_valu_insts"] = v["valu_insts"]
json.dump(out, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["total_bytes"] / 1e6, 2) for k, v in per.items()}, indent=1))
