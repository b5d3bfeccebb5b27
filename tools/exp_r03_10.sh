set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp10.log
: > $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $O
timeout 1500 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
tail -c 400 gpurun_out/r03/bench_default.err >> $O
python - >> $O <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_default.json').read().strip().splitlines()[-1])
for k in ("value","ms_per_step","value_storage","value_h1_lossless_checkpoint","ms_per_step_h1_lossless_checkpoint","split_ms","decode_loop_roofline","decode_loop_roofline_h1_lossless_checkpoint","roofline","kernel_us","codec_roofline","plugin_surface","end_to_end_with_extractor","gpu_over_cpu"):
    print(k, json.dumps(d.get(k)))
print("cpu_baseline", json.dumps({k:v for k,v in d.get("cpu_baseline",{}).items() if k!="reference_algorithm_extrapolated"}))
print("cpu_ref_alg", json.dumps(d.get("cpu_baseline",{}).get("reference_algorithm_extrapolated")))
PY
timeout 900 python bench.py --workload c4 --no-cpu-baseline --no-plugin > gpurun_out/r03/bench_c4.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_c4.json').read().strip().splitlines()[-1])
print('c4', d['value'], d['ms_per_step'], d.get('value_h1_lossless_checkpoint'), d.get('decode_loop_roofline'), d.get('decode_loop_roofline_h1_lossless_checkpoint'))" >> $O
timeout 900 python bench.py --weights fp8 --codec mx8 --batch 16 --no-cpu-baseline --no-plugin > gpurun_out/r03/bench_c5.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_c5.json').read().strip().splitlines()[-1])
print('c5', d['value'], d['ms_per_step'], d.get('split_ms'))" >> $O
timeout 900 python bench.py --batch 16 --no-cpu-baseline --no-plugin --no-second > gpurun_out/r03/bench_b16.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_b16.json').read().strip().splitlines()[-1])
print('b16 h2', d['value'], d['ms_per_step'], d.get('split_ms'))" >> $O
cat $O
