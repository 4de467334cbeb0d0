"""gpurun_out/prof_<tag>/ -> profiles/<round>/<tag>_* and profiles/traffic.json (per-kernel HBM bytes per launch).

usage: python tools/collect_profiles.py <tag> <round dir, e.g. r01>
FETCH_SIZE / WRITE_SIZE are reported in KiB units per dispatch (MI355X_MICROARCH.md, rocprofv3 section); FETCH_SIZE is
left un-doubled for these kernels (4..12-byte per-lane gathers, not the 16-byte/lane streams the gfx950 x2 correction
was calibrated on); WRITE_SIZE is cross-checked on k_pack_frame, whose output is exactly rows*cols*4 bytes.
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
per = {k: {"fetch_bytes": fetch.get(k, 0.0), "write_bytes": write.get(k, 0.0),
           "total_bytes": fetch.get(k, 0.0) + write.get(k, 0.0)} for k in sorted(set(fetch) | set(write)) if "pba::" in k}
out = {
    "unit": "bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB x 1024 from separate rocprofv3 --pmc passes; FETCH_SIZE NOT "
            "doubled (per-lane gathers, not the 16-byte/lane streams the gfx950 x2 correction was calibrated on); WRITE_SIZE "
            "cross-checked on k_pack_frame (1.87 MB known)",
    "workload": "configs[1]: 8 frames x 50k points, R=2, 400k observations",
    "source": "%s/%s_pmc_*.csv" % (dst, tag),
    "per_kernel": per,
}
for k, v in per.items():
    if "k_sample<2, true, 4, true, true>" in k:
        out["k_sample<JAC>"] = v["total_bytes"]
    if k.endswith("k_schur"):
        out["k_schur"] = v["total_bytes"]
json.dump(out, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["total_bytes"] / 1e6, 2) for k, v in per.items()}, indent=1))
