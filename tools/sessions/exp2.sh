#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp2_v2.txt
echo "# v2 kernel: correctness vs v1, then timings" > $O
timeout 900 python3 tools/v2_check.py 20 24 >> $O 2>&1
echo "rc=$?" >> $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1,"DNM_SWZ":16}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":0}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":2,"DNM_TILE_DMA":1}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":2,"DNM_TILE_DMA":1}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":2,"DNM_TILE_DMA":0}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":17,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":15,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}},
{"B":12,"R":3,"mode":2,"amin":-1,"g":9,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":0,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}},
{"B":12,"R":3,"mode":2,"amin":-1,"g":9,"cp":98,"env":{"DNM_KERNEL":1,"DNM_SWZ":0}},
{"B":12,"R":4,"mode":2,"amin":-1,"g":9,"cp":98,"env":{"DNM_KERNEL":1,"DNM_SWZ":0}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
timeout 1500 bash tools/prof_multi.sh 30 '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":1}}' '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_SWZ":16,"DNM_GATHER_INFLIGHT":1,"DNM_TILE_DMA":0}}' >> $O 2>&1
tail -5 $O
