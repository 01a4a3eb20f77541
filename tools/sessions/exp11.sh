#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp12_addr.txt
echo "# after the XOR-linear gather addressing" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10],"nodiag":1},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10]},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10,11]},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10,11,12]},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10,11,12,13,14]},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16]},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16],"nodiag":1},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,1,2,3,4,5],"nodiag":1},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0],"nodiag":1},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[12,13,14,15,16],"nodiag":1},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"bonds":[0,12,13,14,15,16],"nodiag":1}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
timeout 900 python3 tools/v2_check.py 20 2>&1 | tail -2 >> $O
