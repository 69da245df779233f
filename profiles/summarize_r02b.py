"""gpurun_out/prof_r02b (profiles/collect_r02.sh) -> profiles/r02_bwd_kernel_stats.csv, r02_bwd_pmc_summary.json,
r02_fp16_frame_kernel_stats.csv"""
import csv, glob, json, os, shutil
from collections import defaultdict


def newest(pattern):
    """gpurun merges every run into gpurun_out/: take the latest file of a kind"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r02b")
out = os.path.join(root, "profiles")
shutil.copy(newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(out, "r02_bwd_kernel_stats.csv"))
shutil.copy(newest(os.path.join(src, "fp16", "*", "*_kernel_stats.csv"))[0], os.path.join(out, "r02_fp16_frame_kernel_stats.csv"))
# FETCH_SIZE correction measured in the same round on k_camera_fwd (profiles/r02_pmc_summary.json)
corr = json.load(open(os.path.join(out, "r02_pmc_summary.json")))
ff, wf = corr["FETCH_SIZE_correction"], corr["WRITE_SIZE_correction"]
res = defaultdict(dict)
for sub, name, f in (("pmc_fetch", "FETCH_SIZE", ff), ("pmc_write", "WRITE_SIZE", wf)):
    acc = defaultdict(list)
    for path in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("<")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k.startswith("rm::"):
            res[k][name + "_bytes_per_launch"] = sum(v) / len(v) * 1024 * f
for k in res:
    res[k]["hbm_bytes_per_launch"] = sum(res[k].values())
rays, S = 512 * 512, 64
summary = {"config": "closed make_test_scene 512x512x64 training step (profiles/bwd_probe.py)",
           "FETCH_SIZE_correction": ff, "WRITE_SIZE_correction": wf, "kernels": res,
           "algorithmic": {"k_render_fwd (recording)": rays * (24 + 12 + 12 + 12 * S), "backward kernels": rays * 12 * (S + 3)}}
json.dump(summary, open(os.path.join(out, "r02_bwd_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))


# ---- ray regeneration (prof_r02c) ----------------------------------------------------------------------------------
srcc = os.path.join(root, "gpurun_out", "prof_r02c")
if newest(os.path.join(srcc, "trace", "*", "*_kernel_stats.csv")):
    shutil.copy(newest(os.path.join(srcc, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(out, "r02_regen_kernel_stats.csv"))
    shutil.copy(newest(os.path.join(srcc, "auto", "*", "*_kernel_stats.csv"))[0], os.path.join(out, "r02_auto_kernel_stats.csv"))
    reg = defaultdict(dict)
    for sub, names, f in (("pmc_fetch", ("FETCH_SIZE",), ff), ("pmc_write", ("WRITE_SIZE",), wf),
                          ("pmc_sq", ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"), None)):
        acc = defaultdict(list)
        for path in newest(os.path.join(srcc, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] in names:
                    acc[(r["Kernel_Name"].split("<")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, name), v in acc.items():
            if k.startswith("rm::k_march_regen") or k.startswith("rm::k_render_finish"):
                m = sum(v) / len(v)
                reg[k][name + ("_bytes_per_launch" if f else "")] = m * 1024 * f if f else m
    rays = 1920 * 1080
    for k, v in reg.items():
        if "FETCH_SIZE_bytes_per_launch" in v and "WRITE_SIZE_bytes_per_launch" in v:
            v["hbm_bytes_per_launch"] = v["FETCH_SIZE_bytes_per_launch"] + v["WRITE_SIZE_bytes_per_launch"]
        if "GRBM_GUI_ACTIVE" in v and "SQ_INSTS_VALU" in v:
            cycles = v["GRBM_GUI_ACTIVE"] / 8
            v["valu_wave_instructions_per_simd_cycle"] = v["SQ_INSTS_VALU"] / 1024 / cycles
    summ = {"config": "make_test_scene2 1920x1080x128, camera (0,0,1), RenderLoop(regen=True) (profiles/regen_driver.py)",
            "FETCH_SIZE_correction": ff, "WRITE_SIZE_correction": wf, "kernels": reg,
            "algorithmic_bytes_per_launch": {"rm::k_march_regen": rays * (24 + 12 + 4), "rm::k_render_finish": rays * (24 + 12 + 12)}}
    json.dump(summ, open(os.path.join(out, "r02_regen_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))
for name in ("r02_regen_probe.txt", "r02_regen_stale_probe.txt"):
    p = os.path.join(root, "gpurun_out", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, name))
