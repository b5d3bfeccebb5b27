"""Per-launch durations of the codec kernels of the LAST decode in a rocprofv3 kernel trace
(gpurun_out/prof_<tag>/stats/**/_kernel_trace.csv written by tools/profile_round.sh)."""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
f = sorted(glob.glob(f"gpurun_out/prof_{tag}/stats/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
NAMES = ("conv_pair_kernel", "conv_mx8_kernel", "conv_out_kernel", "from_codes_kernel", "conv_mfma_kernel", "enc_conv_in_kernel", "rvq_stage_kernel")
conv = [r for r in rows if any(n in r["Kernel_Name"] for n in NAMES)]
conv.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(conv) if "from_codes" in r["Kernel_Name"]]
seg = conv[idx[-1]:]
tot = 0.0
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f'{r["Kernel_Name"].split("(")[0].replace("void ", ""):24s} grid {r["Grid_Size_X"]:>8s} x {r["Grid_Size_Y"]:>5s} x {r["Grid_Size_Z"]:>4s}  {d:8.1f} us')
print(f"total {tot / 1e3:.2f} ms over {len(seg)} launches")
