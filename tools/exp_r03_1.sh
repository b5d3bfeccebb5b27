set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp1.log
: > $O
LIB=vaura_amd/csrc/libvaura_hip.so
echo "== C++ driver: 1 chain x 16 rows (baseline)" >> $O
timeout 300 tools/pmc_driver $LIB --time 5 >> $O 2>&1
echo "== C++ driver: 2 chains x 8 rows, thread per chain" >> $O
timeout 300 tools/pmc_driver $LIB --chains 2 --time 5 >> $O 2>&1
echo "== C++ driver: 2 chains x 8 rows, one host thread" >> $O
PMC_ONE_THREAD=1 timeout 300 tools/pmc_driver $LIB --chains 2 --time 5 >> $O 2>&1
echo "== C++ driver: 4 chains x 4 rows, thread per chain" >> $O
timeout 300 tools/pmc_driver $LIB --chains 4 --time 5 >> $O 2>&1
echo "== C++ driver: 1 chain x 8 rows alone" >> $O
timeout 300 tools/pmc_driver $LIB --rows 8 --time 5 >> $O 2>&1
echo "== python host enqueue: default env" >> $O
timeout 600 python tools/time_loop_parts.py >> $O 2>&1
echo "== python host enqueue: DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" >> $O
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 600 python tools/time_loop_parts.py >> $O 2>&1
echo "== python host enqueue: LD_PRELOAD rocm 7.2 runtime" >> $O
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 timeout 600 python tools/time_loop_parts.py >> $O 2>&1
echo "== python host enqueue: LD_PRELOAD rocm 7.2 runtime + hsa" >> $O
LD_PRELOAD="/opt/rocm/lib/libhsa-runtime64.so.1 /opt/rocm/lib/libamdhip64.so.7" timeout 600 python tools/time_loop_parts.py >> $O 2>&1
cat $O
