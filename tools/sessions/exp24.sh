#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp24_tuning.txt
echo "# plan / cache-policy tuning under the swizzled layout (same process, same box)" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":102},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":34},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":96},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":66},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":99},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":32},
{"B":12,"R":3,"mode":2,"amin":5,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":3,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":7,"cp":98},
{"B":12,"R":3,"mode":2,"amin":6,"g":7,"cp":98},
{"B":11,"R":3,"mode":2,"amin":4,"g":7,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
