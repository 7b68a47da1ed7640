"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/.
Usage: tools/profile_collect.py <tag> [<name under profiles/> mode]   ("mode": a tools/profile_mode.sh directory of another bench mode — not copied to latest_pmc.json)

  profiles/<tag>_bench.json            the bench line
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the same command
  profiles/<tag>_pmc.json              per-kernel mean counters per launch + derived HBM traffic
"""
import csv, glob, json, os, shutil, sys, collections
src = f"gpurun_out/{sys.argv[1]}"
tag = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
mode_only = len(sys.argv) > 3 and sys.argv[3] == "mode"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"{src}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("awsm::"):
                ctr[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, c in sorted(ctr.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["launches_sampled"] = max(len(v) for v in c.values())
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide
        # (16 B/lane) streaming reads -> doubled there; other access widths are uncalibrated, so both readings are kept.
        m["hbm_read_bytes_raw"] = m["FETCH_SIZE"] * 1024.0
        m["hbm_read_bytes_wide_corrected"] = m["FETCH_SIZE"] * 2048.0
        m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024.0
        m["hbm_traffic_bytes"] = m["hbm_read_bytes_wide_corrected"] + m["hbm_write_bytes"]
    out[k] = m
bench = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
doc = {"tag": tag, "workload": bench["config"], "kernels": out}
# Static VALU mix of the kernels the roofline talks about, from the ISA of the sources as they are now (hipcc -S here; no GPU needed),
# weighted by the issue classes tools/valu_probe.hip measured on gfx950: full rate 1 (= 2 cycles per wave64 instruction: the part's
# 157 TFLOP/s of vector FP32 are 32 FMA lanes per SIMD and clock), half rate 1.8, transcendental 3.5 (tools/isa_cost.py).
try:
    if mode_only: raise RuntimeError("mode profile")
    import subprocess, tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_cost
    mix = {}
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "awsm-renderer_amd", "csrc")
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function --cuda-device-only -S".split()
    for src, kernels in (("kernels_shade.hip", {"awsm::k_shade_lean<false, 0, false>": "k_shade_leanILb0ELi0ELb0", "awsm::k_shade<0>": "k_shadeILi0"}),
                         ("kernels_geometry.hip", {"awsm::k_raster_tile<1>": "k_raster_tileILi1"})):
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", asm, src], cwd=csrc, check=True, capture_output=True)
            for pretty, mangled in kernels.items():
                n, cost, cls, slowed, _ = isa_cost.analyse(asm, mangled)
                mix[pretty] = {"static_valu_instructions": n, "weighted_cost": cost, "avg_weight": cost / n, "cycles_per_wave64_instruction": 2.0 * cost / n,
                               "full": cls["full"], "half": cls["half"], "transcendental": cls["trans"], "full_rate_ops_slowed_by_sgpr_or_literal": slowed}
    doc["valu_mix"] = mix
except Exception as e:      # no compiler here: the bench falls back to 4 cycles per instruction
    print("valu_mix not computed:", e)
json.dump(doc, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
if not mode_only: shutil.copy(f"profiles/{tag}_pmc.json", "profiles/latest_pmc.json")
print(json.dumps({k: {n: v[n] for n in ("hbm_traffic_bytes", "FETCH_SIZE", "WRITE_SIZE") if n in v} for k, v in out.items()}, indent=1))
