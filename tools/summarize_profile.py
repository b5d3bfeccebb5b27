"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small, committed summaries under profiles/:

  <tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the python bench command
  <tag>_bench_under_rocprof.json      the JSON line that run printed
  <tag>_driver_kernel_stats_<w>.csv   the same kernels through tools/pmc_driver (the product .so), w = h2 | h1 (fp16 planes per weight)
  <tag>_pmc_hbm_bytes_<w>.json        HBM bytes per launch per kernel from separate FETCH_SIZE / WRITE_SIZE passes on the driver
  <tag>_codec_mfma.json, <tag>_avclip_mfma.json, <tag>_prefill_h{1,2}_mfma.json   MFMA / LDS / HBM counters of the MFMA-bound stages
                                                  through tools/mfma_driver
  <tag>_stage_stamps_<mode>_<w>.json  per-phase decomposition of the decode-step stages (in-kernel s_memrealtime stamps)
  kernel_names.json                   storage | "c4" -> decode-step stage -> kernel name as rocprofv3 prints it (bench.py quotes it)

The summariser REFUSES to write the PMC record unless every decode-step kernel name of the PMC passes also appears in the
bench's own kernel stats: counters from an old build or another code object cannot end up next to fresh timings.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = f"gpurun_out/prof_{tag}"
# second argument: where the summaries go (default profiles/).  tools/profile_round.sh runs this ON the GPU box into gpurun_out/profiles_<tag>/
# — the per-dispatch counter tables of 228-step passes are too large to travel back — and the builder copies that directory into profiles/.
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
os.makedirs(DST, exist_ok=True)


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def stage_of(name):
    """decode-step stage of a kernel name (template arguments: G, NW, T, EPI, NORM, ...)."""
    if name.startswith("attention_step256_kernel") or name.startswith("attention_split_kernel") or name.startswith("attention_step_kernel"):
        return "attn"
    m = re.match(r"gemv3_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        g, epi, norm = int(m.group(1)), int(m.group(4)), m.group(5) == "true"
        if epi == 0 and norm:
            return "qkv"
        if epi == 1:
            return "w2" if g == 16 else "wo"
        if epi == 2:
            return "w13"
        if epi == 4:
            return "heads"
    m = re.match(r"mlp_engine_kernel<(\d+), (true|false)", name)   # <WT, QKV, RBK, ATT>: w1||w3 -> w2 (-> next layer's qkv) in one launch
    if m:
        return "mlp" if m.group(2) == "true" else "mlp_last"
    m = re.match(r"gemv3h_kernel<(\d+), ", name)      # row-split pair kernels: G2 = k-group pairs per wave
    if m:
        return "w2" if int(m.group(1)) == 8 else "wo"
    if name.startswith("sample_kernel"):
        return "sample"
    if name.startswith("embed_kernel"):
        return "embed"
    return None


def clean(name):
    return re.sub(r"\(.*$", "", name.replace("void ", "")).strip()


def write_stats(path, dst, header):
    rows = list(csv.DictReader(open(path)))
    with open(dst, "w") as f:
        f.write(f"# {header}\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    return rows


stats = newest(f"{src}/stats/**/*kernel_stats.csv")
rows, bench_names = [], set()
if stats:        # (absent when only the MFMA part of the round ran on this box: tools/profile_round.sh <tag> mfma)
    rows = write_stats(stats, f"{DST}/{tag}_bench_kernel_stats.csv",
                       "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second --no-plugin --no-extra-configs   (auto -> h2 storage, un-rounded checkpoint)")
    bench_names = {clean(r["Name"]) for r in rows}
    log = [l for l in open(f"{src}/bench_stats.log").read().splitlines() if l.startswith("{")]
    open(f"{DST}/{tag}_bench_under_rocprof.json", "w").write((log[-1] if log else "{}") + "\n")


def per_kernel(path, counter):
    f = newest(f"{path}/**/*counter_collection.csv")
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[clean(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


names = {}
for w in ("h2", "h1", "h2_rows32", "fp8h_rows32"):
    st = newest(f"{src}/drv_stats_{w}/**/*kernel_stats.csv")
    if not st:
        continue
    rows_n = 32 if w.endswith("rows32") else 16
    wname = w.split("_")[0]
    kvflag = " --kv f16" if wname == "fp8h" else ""
    drows = write_stats(st, f"{DST}/{tag}_driver_kernel_stats_{w}.csv",
                        f"rocprofv3 --kernel-trace --stats -- tools/pmc_driver vaura_amd/csrc/libvaura_hip.so --weights {wname}{kvflag} --rows {rows_n} --steps 24 --pos0 10 --stride 9")
    avg_ns = {clean(r["Name"]): float(r["AverageNs"]) for r in drows}
    names[w] = {stage_of(n): n for n in avg_ns if stage_of(n)}
    fetch = per_kernel(f"{src}/drv_FETCH_SIZE_{w}", "FETCH_SIZE")
    write = per_kernel(f"{src}/drv_WRITE_SIZE_{w}", "WRITE_SIZE")
    step_kernels = {k for k in fetch if stage_of(k)}
    if w == "h2" and bench_names:
        missing = sorted(step_kernels - bench_names)
        if missing:
            raise SystemExit(f"PMC kernel names not in the bench's kernel stats (stale build?): {missing}")
    out = {"weights": wname, "rows": rows_n, "kv_cache": "f16" if kvflag else "f32", "kernels": {},
           "source": f"tools/profile_round.sh {tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- tools/pmc_driver libvaura_hip.so "
                     f"--weights {wname}{kvflag} --rows {rows_n} --steps 24 --pos0 10 --stride 9 (24 cache lengths sampled over the WHOLE loop: 10, 19, ..., 217, mean 113.5 = the loop's mean)",
           "formula": "hbm_bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (MI355X_MICROARCH.md §HBM: FETCH_SIZE counts half the "
                      "bytes of wide coalesced reads on gfx950; WRITE_SIZE is exact)"}
    for k in sorted(step_kernels):
        fs, n = fetch[k]
        ws = write.get(k, (0.0, 0))[0]
        out["kernels"][k] = {"stage": stage_of(k), "launches": n, "FETCH_SIZE_KiB_raw": fs, "WRITE_SIZE_KiB_raw": ws,
                             "hbm_read_bytes_per_launch": fs * 2048, "hbm_write_bytes_per_launch": ws * 1024,
                             "hbm_bytes_per_launch": fs * 2048 + ws * 1024, "avg_ns_same_driver_run": avg_ns.get(k)}
    json.dump(out, open(f"{DST}/{tag}_pmc_hbm_bytes_{w}.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{w} {v['stage']:6s} {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB  {(v['avg_ns_same_driver_run'] or 0) / 1e3:7.2f} us  {k[:80]}")
stats_c4 = newest(f"{src}/stats_c4/**/*kernel_stats.csv")
if stats_c4:
    crow = write_stats(stats_c4, f"{DST}/{tag}_c4_kernel_stats.csv",
                       "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-second --no-plugin")
    c4 = {}
    for r in sorted(crow, key=lambda r: -float(r["TotalDurationNs"])):       # per stage: the kernel with the largest total time
        st_ = stage_of(clean(r["Name"]))
        if st_ and st_ not in c4:
            c4[st_] = clean(r["Name"])
    names["c4"] = c4
if names:
    json.dump(names, open(f"{DST}/kernel_names.json", "w"), indent=1)

# ---- MFMA-bound stages (tools/mfma_driver): time, MFMA work and busy cycles, LDS conflicts, HBM bytes per kernel
def counters(path):
    f = newest(f"{path}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[clean(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


for m in ("codec", "avclip", "prefill_h2", "prefill_h1"):
    st = newest(f"{src}/mfma_stats_{m}/**/*kernel_stats.csv")
    if not st:
        continue
    mrows = write_stats(st, f"{DST}/{tag}_{m}_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- tools/mfma_driver libvaura_hip.so {m} 8   (3 passes of the stage)")
    A, Bc = counters(f"{src}/mfma_pmcA_{m}"), counters(f"{src}/mfma_pmcB_{m}")
    Fz, Wz = counters(f"{src}/mfma_FETCH_SIZE_{m}"), counters(f"{src}/mfma_WRITE_SIZE_{m}")
    out = {"stage": m, "clips": 8, "source": f"tools/profile_round.sh {tag}: rocprofv3 --kernel-trace --pmc ... -- tools/mfma_driver libvaura_hip.so {m} 8 "
                                             "(separate passes A: MFMA ops / busy cycles, B: LDS + MFMA instruction counts, FETCH_SIZE, WRITE_SIZE; averages per launch)",
           "derived": {"mfma_flops": "SQ_INSTS_VALU_MFMA_MOPS_F16 (or _F32) x 512 (the counter's unit)",
                       "elapsed_cycles": "GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)",
                       "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (elapsed_cycles x 1024 SIMDs)",
                       "mfma_tflops": "mfma_flops / AverageNs of the kernel-trace pass of the same command",
                       "frac_of_dense_fp16_peak": "mfma_tflops / 2500 (MI355X_MICROARCH.md: ~2.5 PFLOP/s dense)",
                       "lds_conflict_frac": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
                       "hbm_bytes": "2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024"},
           "kernels": {}}
    tot_ns = sum(float(r["TotalDurationNs"]) for r in mrows)
    for r in sorted(mrows, key=lambda r: -float(r["TotalDurationNs"])):
        k = clean(r["Name"])
        a, b2 = A.get(k, {}), Bc.get(k, {})
        if float(r["TotalDurationNs"]) < 0.005 * tot_ns:
            continue
        mops = a.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        avg_ns = float(r["AverageNs"])
        rec = {"calls": int(r["Calls"]), "avg_us": round(avg_ns / 1e3, 2), "share_of_stage_time": round(float(r["TotalDurationNs"]) / tot_ns, 4),
               "raw": {**{c: a[c] for c in sorted(a)}, **{c: b2[c] for c in sorted(b2)}}}
        if mops and avg_ns:
            rec["mfma_tflops"] = round(mops * 512 / avg_ns / 1e3, 1)
            rec["frac_of_dense_fp16_peak"] = round(mops * 512 / avg_ns / 1e3 / 2500.0, 4)
        if cyc and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            rec["mfma_busy_frac"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
            rec["clock_ghz_from_gui_active"] = round(cyc / avg_ns, 3)
        if b2.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_frac"] = round(b2.get("SQ_LDS_BANK_CONFLICT", 0.0) / b2["SQ_LDS_IDX_ACTIVE"], 4)
        if k in Fz:
            rec["hbm_bytes_per_launch"] = Fz[k].get("FETCH_SIZE", 0.0) * 2048 + Wz.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        out["kernels"][k] = rec
        print(f"{m:6s} {rec['avg_us']:9.2f} us x{rec['calls']:4d} {rec.get('mfma_tflops', 0):7.1f} TF busy {rec.get('mfma_busy_frac', 0):.3f} lds-conf {rec.get('lds_conflict_frac', 0):.3f}  {k[:70]}")
    json.dump(out, open(f"{DST}/{tag}_{m}_mfma.json", "w"), indent=1)

for f in glob.glob(f"{src}/stamps_*.json"):
    base = os.path.basename(f).replace("stamps_", "")
    open(f"{DST}/{tag}_stage_stamps_{base}", "w").write(open(f).read())
for r in rows[:14]:
    print(clean(r["Name"])[:70].ljust(70), r["Calls"].rjust(7), "%9.2f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
