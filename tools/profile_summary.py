#!/usr/bin/env python3
"""Condense a tools/profile_round.sh run into profiles/<tag>_kernel_stats.csv and profiles/<tag>_traffic.json."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
variant = sys.argv[3] if len(sys.argv) > 3 else ""      # "" = the headline (writes profiles/traffic.json too); else a label
n_conv = 53
CONV_KEYS = ("conv_mfma", "conv_flatd", "conv_pwr", "conv_c3", "conv_f32", "conv_generic")
def is_conv(name):
    return any(k in name for k in CONV_KEYS) and "prep" not in name
os.makedirs("profiles", exist_ok=True)
# kernel stats
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("profiles/%s_kernel_stats.csv" % tag, "w") as w:
        w.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline %s (7 steps incl. warm-up)\n" % variant)
        w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            n = r["Name"]
            n = n if len(n) <= 110 else n[:107] + "..."
            w.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    conv = [r for r in rows if is_conv(r["Name"])]
    prep = [r for r in rows if "prep" in r["Name"]]
    tot_conv = sum(float(r["TotalDurationNs"]) for r in conv)
    calls_conv = sum(int(r["Calls"]) for r in conv)
    tot_prep = sum(float(r["TotalDurationNs"]) for r in prep)
    steps = calls_conv / 53.0   # 53 conv launches per step (timed + warm-up steps)
    print("conv kernels: %d launches, %.3f ms per step, avg %.1f us per launch; prep: %.3f ms per step" % (
        calls_conv, tot_conv / steps / 1e6, tot_conv / calls_conv / 1e3, tot_prep / steps / 1e6))
def pmc_sum(kind, counter):
    tot, n = 0.0, 0
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % kind, recursive=True):
        for r in csv.DictReader(open(f)):
            if is_conv(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"]); n += 1
    return tot, n
fs, nf = pmc_sum("fetch", "FETCH_SIZE")
ws, nw = pmc_sum("write", "WRITE_SIZE")
if nf and nw:
    # KiB units; gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> doubled
    fetch_b = 2.0 * fs * 1024 / nf
    write_b = ws * 1024 / nw
    sys.path.insert(0, os.getcwd())
    from quantize_amd.build import source_sha16
    d = {"tag": tag, "source_sha16": source_sha16(), "launches_profiled": nf, "fetch_size_kib_per_launch_raw": fs / nf, "write_size_kib_per_launch": ws / nw,
         "fetch_bytes_per_launch_corrected_x2": fetch_b, "write_bytes_per_launch": write_b,
         "hbm_bytes_per_launch": fetch_b + write_b,
         "variant": variant or "headline",
         "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over python3 bench.py --steps 5 --warmup 2 %s; conv_* kernels only (53 per step); " % variant +
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for 16 B/lane streams)"}
    json.dump(d, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    # the file bench.py reads `roofline.traffic` from: the headline's, or the variant's own
    slug = {"": "", "--float-input": "_float_input", "--fused-requant": "_fused_requant", "--w-bits 4 --a-bits 4": "_w4a4",
            "--asymmetric": "_asymmetric"}.get(variant.strip())
    if slug is not None:
        json.dump(d, open("profiles/traffic%s.json" % slug, "w"), indent=1)
    print(json.dumps(d))
