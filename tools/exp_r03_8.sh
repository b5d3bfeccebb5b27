set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp8.log
: > $O
LIB=vaura_amd/csrc/libvaura_hip.so
echo "== DPP / permlane-swap reductions: h1, h2" >> $O
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0 --weights h1 >> $O 2>&1
timeout 600 tools/pmc_driver $LIB --time 5 --flags 0 --weights h2 >> $O 2>&1
timeout 3000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generate.py tests/test_gpu_plugins.py -q -x 2>&1 | tail -15 >> $O
timeout 600 python tools/time_codec.py >> $O 2>&1
cat $O
