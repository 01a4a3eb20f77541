#!/bin/bash
# GPU box: do the exchange and the rank-local kernels of the NATIVE partitioned multiply really run at the same time?
# Kernel trace (start / end stamps) of tools/rccl_loopback_bench.py's rank 0 of 2 at L=31 (RCCL looped back to the rank):
# the time RCCL's kernels and the tile passes are BOTH running, per native multiply.  (In loop-back both are bound by the
# same HBM, so the multiply's time is the sum of the two -- what this shows is that the schedule does not serialise them.)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp DNM_EXPERIMENTAL=1
rm -rf /tmp/rp_ov
rocprofv3 --kernel-trace -f csv -d /tmp/rp_ov -o t -- python3 tools/rccl_loopback_bench.py 31 2 0 > /tmp/rp_ov.log 2>&1; grep "native call" /tmp/rp_ov.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/rp_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
nccl = [(a, b) for a, b, n in ev if "nccl" in n.lower() or "rccl" in n.lower()]
tile = [(a, b, n) for a, b, n in ev if "tile_pass" in n]
print("kernels in the trace: %d RCCL, %d tile passes" % (len(nccl), len(tile)))
def overlap(a, b, c, d):
    return max(0, min(b, d) - max(a, c))
# order of the run: host multiplies (7 RCCL kernels), host exchange alone (7), NATIVE multiplies (7), native exchange alone (7)
for (a, b) in nccl:
    both = sum(overlap(a, b, c, d) for c, d, _ in tile)
    names = sorted({n.split("<")[1].split(">")[0] for c, d, n in tile if overlap(a, b, c, d) > 0})
    print("   RCCL kernel %.3f ms long: tile passes running during %.3f ms of it (%d %%)  %s"
          % ((b - a) / 1e6, both / 1e6, round(100 * both / max(1, b - a)), names))
PY
