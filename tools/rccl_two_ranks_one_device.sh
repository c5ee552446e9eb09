#!/bin/bash
# two ranks of tools/rccl_two_ranks_one_device.py on one device, each under its own timeout
R=${GRAFT_REPO_ROOT:-$(pwd)}
ID=/tmp/dto_uid_$$
rm -f $ID
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 150 python3 $R/tools/rccl_two_ranks_one_device.py 0 2 $ID > $R/gpurun_out/r03b/rank0.log 2>&1 &
P0=$!
timeout -k 10 150 python3 $R/tools/rccl_two_ranks_one_device.py 1 2 $ID > $R/gpurun_out/r03b/rank1.log 2>&1 &
P1=$!
wait $P0; E0=$?
wait $P1; E1=$?
echo "exit codes $E0 $E1"
tail -n 20 $R/gpurun_out/r03b/rank0.log $R/gpurun_out/r03b/rank1.log
