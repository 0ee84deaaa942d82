#!/bin/bash
# rocprofv3 counter passes over bench.py --decode (BASELINE configs[3]: 1 000 PVs, k = 5, 100 positions): HBM-side traffic of the decode
# attention kernel (FETCH_SIZE / WRITE_SIZE, one pass each).  Counters in their own runs, no trace domains, the program directly after `--`.
#   tools/pmc_decode.sh OUTDIR PREFIX     ->  PREFIX_pmc_decode_attn.{json,txt}
set -e
export TMPDIR=/tmp
out=$1; prefix=$2
mkdir -p "$out"
B="python3 bench.py --decode --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "decode_attn" -d "$out/fetch" -o p -f csv -- $B > "$out/fetch.log" 2>&1 || tail -3 "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "decode_attn" -d "$out/write" -o p -f csv -- $B > "$out/write.log" 2>&1 || tail -3 "$out/write.log"
python3 - "$out" "$prefix" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out, prefix = sys.argv[1], sys.argv[2]
def load(sub, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
lines = ["rocprofv3 PMC passes over bench.py --decode --no-cpu-baseline (warm-up + timed + instrumented chunk of 1 000 molecules x 5 beams x 100 positions); KiB per launch, raw counters",
         f"{'kernel':60s} {'n':>6s} {'FETCH_SIZE':>12s} {'WRITE_SIZE':>12s}"]
tf = tw = n = 0
for k in sorted(fe):
    f, w = fe[k], wr.get(k, [])
    if not w:
        continue
    lines.append(f"{k:60s} {len(f):6d} {sum(f) / len(f):12.1f} {sum(w) / len(w):12.1f}")
    tf += sum(f); tw += sum(w) * len(f) / len(w); n += len(f)
js = {"workload": {"molecules": 1000, "beams": 5, "positions": 100, "chunk": 1000}, "kernel": "decode_attn_* (self- and cross-attention launches over the K/V cache)",
      "command": "tools/pmc_decode.sh: rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-include-regex decode_attn -- python3 bench.py --decode --no-cpu-baseline (one pass per counter)",
      "launches_counted": n, "fetch_bytes_per_launch": 2 * tf * 1024 / n, "write_bytes_per_launch": tw * 1024 / n, "traffic_bytes_per_launch": (2 * tf + tw) * 1024 / n,
      "corrections": "counter unit KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; "
                     "Infinity-Cache hits are counted, so this is L2<->fabric traffic, an upper bound on HBM bytes"}
json.dump(js, open(prefix + "_pmc_decode_attn.json", "w"), indent=1)
open(prefix + "_pmc_decode_attn.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines)); print(json.dumps(js)[:600])
PY
