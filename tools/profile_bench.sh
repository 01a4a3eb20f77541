#!/bin/bash
# GPU box: rocprofv3 summaries of the default bench.py run for profiles/.
#   1. --kernel-trace --stats      (per-kernel durations)
#   2. --pmc FETCH_SIZE            (separate pass, as the MI355X guide prescribes)
#   3. --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# Results: gpurun_out/profiles/{kernel_stats.csv,kernel_trace_tail.csv,pmc_summary.json,bench_line.json}
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/profiles; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp_trace -o t -- $CMD > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
find /tmp/rp_trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
TR=$(find /tmp/rp_trace -name "*kernel_trace.csv" | head -1)
if [ -n "$TR" ]; then head -1 $TR > $OUT/kernel_trace_tail.csv; grep tile_pass $TR | tail -24 >> $OUT/kernel_trace_tail.csv; fi
rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/rp_pmc1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d /tmp/rp_pmc2 -o p -- $CMD > /dev/null 2>&1
# BASELINE configs[1] (L=26 XXZ): counter bytes of its multiply for bench.py's secondary.multiply_L26_xxz
CMD2="python3 bench.py --L 26 --model xxz --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
$CMD2 > $OUT/bench_line_L26.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/rp_pmc3 -o p -- $CMD2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d /tmp/rp_pmc4 -o p -- $CMD2 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
sys.path.insert(0, ".")
import bench
KHASH = bench.kernel_source_hash()
line = json.loads(open(out + "/bench_line.json").read().strip().splitlines()[-1])
L, launches = line["config"]["L"], line["config"]["launches_per_step"]
dim = 1 << L
def load(pattern):
    f = glob.glob(pattern, recursive=True)[0]
    return list(csv.DictReader(open(f)))
def per_dispatch(rows):
    d = {}
    for r in rows:
        if "tile_pass" not in r["Kernel_Name"]: continue
        d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return d
d1 = per_dispatch(load("/tmp/rp_pmc1/**/*counter_collection.csv"))
d2 = per_dispatch(load("/tmp/rp_pmc2/**/*counter_collection.csv"))
ids1, ids2 = sorted(d1)[-launches:], sorted(d2)[-launches:]
passes = []
for a, b in zip(ids1, ids2):
    fetch = d1[a]["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of a 16 B/lane coalesced stream
    write = d2[b]["WRITE_SIZE"] * 1024
    hit, miss = d2[b].get("TCC_HIT_sum", 0), d2[b].get("TCC_MISS_sum", 0)
    passes.append({"fetch_bytes": fetch, "write_bytes": write, "l2_hit_rate": hit / max(1.0, hit + miss)})
tot = sum(p["fetch_bytes"] + p["write_bytes"] for p in passes)
# config 2
line2 = json.loads(open(out + "/bench_line_L26.json").read().strip().splitlines()[-1])
l2 = line2["config"]["launches_per_step"]
d3 = per_dispatch(load("/tmp/rp_pmc3/**/*counter_collection.csv"))
d4 = per_dispatch(load("/tmp/rp_pmc4/**/*counter_collection.csv"))
tot2 = sum(d3[a]["FETCH_SIZE"] * 1024 * 2 for a in sorted(d3)[-l2:]) + sum(d4[b]["WRITE_SIZE"] * 1024 for b in sorted(d4)[-l2:])
config2 = {"L": 26, "model": "xxz", "plan_signature": line2["config"]["plan_signature"], "kernel_source_hash": KHASH,
           "launches_per_step": l2, "hbm_bytes_per_step": tot2, "bytes_per_amplitude": tot2 / (1 << 26),
           "ms_per_step": line2["ms_per_step"]}
summ = {"L": L, "n_gpus": 1, "plan": str(line["config"].get("plan_mode", 2)),
        "plan_signature": line["config"].get("plan_signature"), "kernel_source_hash": KHASH, "config2": config2,
        "launches_per_step": launches,
        "passes": passes, "hbm_bytes_per_step": tot, "hbm_bytes_per_launch": tot / launches,
        "bytes_per_amplitude": tot / dim,
        "note": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane streams, calibrated on mdot: 2 x 16 GiB read "
                "reports 17.18 GB); WRITE_SIZE as reported; Infinity-Cache hits are included in FETCH_SIZE",
        "bench_line": line}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps({k: summ[k] for k in ("launches_per_step", "bytes_per_amplitude", "hbm_bytes_per_launch", "kernel_source_hash")}))
print(json.dumps(config2))
for p in passes: print(p)
PY
cat $OUT/kernel_stats.csv | head -12
