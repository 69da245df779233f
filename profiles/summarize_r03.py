"""gpurun_out/prof_<tag>/ (profiles/collect_r03.sh) -> profiles/<tag>_bwd_kernel_stats_{512,2048}.csv,
profiles/<tag>_bwd_pmc_summary.json, profiles/<tag>_config5_kernel_stats.csv, profiles/<tag>_config5_pmc_summary.json and
profiles/traffic_bwd.json (read by bench.py for fwd_bwd.roofline.traffic; stamped with the hash of the kernel sources).

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB,
corrected by the factor measured IN THE SAME PASS on k_camera_fwd (known byte count, the same 12-byte-per-lane
pattern; FETCH_SIZE reports half of a coalesced stream on gfx950)."""
import csv
import glob
import importlib.util
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
out = os.path.join(root, "profiles")


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


def short(name):
    name = name.replace("void ", "")
    base = name.split("<")[0].split("(")[0]
    if "k_render_bwd" in name:            # keep the shader kind of the instantiation
        kind = name.rsplit(",", 1)[-1].split(">")[0].strip() if "," in name else "?"
        base += f"<kind {kind}>"
    return base.replace("rm::", "")


def counters(sub):
    agg = defaultdict(list)
    for f in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def durations(sub):
    res = {}
    for f in newest(os.path.join(src, sub, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            res[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return res


def section(prefix, cal_bytes, kernels_like):
    fetch, _ = counters(prefix + "_fetch")
    write, _ = counters(prefix + "_write")
    sq, _ = counters(prefix + "_sq")
    dur = durations(prefix + "_trace")
    ff = cal_bytes / (fetch[("k_camera_fwd", "FETCH_SIZE")] * 1024)
    wf = cal_bytes / (write[("k_camera_fwd", "WRITE_SIZE")] * 1024)
    ks = {}
    for (k, name), v in sorted(fetch.items()):
        if name != "FETCH_SIZE" or not any(s in k for s in kernels_like):
            continue
        e = ks.setdefault(k, {})
        e["read_bytes_per_launch"] = v * 1024 * ff
        e["written_bytes_per_launch"] = write.get((k, "WRITE_SIZE"), 0.0) * 1024 * wf
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["written_bytes_per_launch"]
        for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if (k, c) in sq:
                e[c] = sq[(k, c)]
        if k in dur:
            e["avg_us"], e["calls_in_trace"] = dur[k]["avg_us"], dur[k]["calls"]
            e["hbm_GBps"] = e["hbm_bytes_per_launch"] / e["avg_us"] / 1e3
        if "SQ_INSTS_VALU" in e and "avg_us" in e:
            # wave-instructions per SIMD per cycle of the launch (1024 SIMDs, 2.4 GHz): ~0.25 is what a mix of
            # 3-operand FMAs / compares / selects can issue (profiles/r03_valu_issue_bench.txt)
            e["valu_wave_instr_per_simd_cycle"] = e["SQ_INSTS_VALU"] / 1024 / (e["avg_us"] * 1e-6 * 2.4e9)
            e["simd_cycles_per_valu_instr"] = 1.0 / e["valu_wave_instr_per_simd_cycle"]
        if "SQ_WAVE_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
            e["mean_resident_waves_per_busy_cycle"] = e["SQ_WAVE_CYCLES"] / e["SQ_BUSY_CYCLES"]
    return {"FETCH_SIZE_correction": ff, "WRITE_SIZE_correction": wf, "kernels": ks}


summary = {"tag": tag, "calibration_kernel": "k_camera_fwd (2 x rays x 12 B read, the same written), in every pass"}
S = 64
for size in (512, 2048):
    if not newest(os.path.join(src, f"bwd{size}_fetch", "*", "*_counter_collection.csv")):
        continue
    rays = size * size
    sec = section(f"bwd{size}", 2 * rays * 12, ("k_render_fwd", "k_render_bwd", "k_bwd_hard", "k_reduce", "k_finish", "k_minmax_init", "k_order"))
    sec["config"] = f"closed make_test_scene {size}x{size}x{S} training step (profiles/train_driver.py)"
    sec["algorithmic_bytes"] = {"k_render_fwd (recording)": rays * (24 + 12 + 12 + 12 * S), "backward kernels": rays * 12 * (S + 3)}
    tot = sum(k["hbm_bytes_per_launch"] for n, k in sec["kernels"].items() if "camera" not in n)
    sec["hbm_bytes_per_step"] = tot
    sec["gpu_us_per_step"] = sum(k.get("avg_us", 0.0) for n, k in sec["kernels"].items() if "camera" not in n)
    summary[f"bwd_{size}"] = sec
    for f in newest(os.path.join(src, f"bwd{size}_trace", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, f"{tag}_bwd_kernel_stats_{size}.csv"))
json.dump(summary, open(os.path.join(out, f"{tag}_bwd_pmc_summary.json"), "w"), indent=1)

spec = importlib.util.spec_from_file_location("_rm_build", os.path.join(root, "ray_marching_amd", "_build.py"))
_build = importlib.util.module_from_spec(spec); spec.loader.exec_module(_build)
traffic = {"source": f"profiles/{tag}_bwd_pmc_summary.json", "sources_hash": _build.sources_hash()}
for size in (512, 2048):
    if f"bwd_{size}" in summary:
        traffic[f"hbm_bytes_per_step_{size}"] = summary[f"bwd_{size}"]["hbm_bytes_per_step"]
json.dump(traffic, open(os.path.join(out, "traffic_bwd.json"), "w"), indent=1)

if newest(os.path.join(src, "c5_fetch", "*", "*_counter_collection.csv")):
    rays = 7680 * 540
    c5 = section("c5", 2 * 1920 * 1080 * 12, ("k_render_fwd", "k_march_regen", "k_render_finish"))
    c5["config"] = "config 5: 32-primitive smooth union, one 7680x540 band, 256 steps, normal shader (profiles/config5_driver.py)"
    c5["algorithmic_bytes_per_launch"] = rays * 36
    f64, _ = counters("c5_f64")
    for k, e in c5["kernels"].items():
        parts = {c: f64.get((k, c)) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")}
        if all(v is not None for v in parts.values()) and f64.get((k, "SQ_INSTS_VALU")):
            e["fp64_instructions"] = parts
            e["fp64_share_of_valu_instructions"] = sum(parts.values()) / f64[(k, "SQ_INSTS_VALU")]
    for f in newest(os.path.join(src, "c5_trace", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, f"{tag}_config5_kernel_stats.csv"))
    json.dump(c5, open(os.path.join(out, f"{tag}_config5_pmc_summary.json"), "w"), indent=1)
    summary["config5"] = c5
print(json.dumps(summary, indent=1))
