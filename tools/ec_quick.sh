#!/bin/bash
# The curve legs of bench.py alone (P-256 CCPoS at width 3 + the operation_length fit over P-256), a few numbers per line.
# usage (GPU box): bash tools/ec_quick.sh <tag>
tag=${1:-ec_quick}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements 0 --ec-elements 1000000 --ccpos-elements 0 \
    --decrypt-elements 0 --no-e2e --skip-cpu --no-shapes > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -5 gpurun_out/$tag.err; exit 1; }
python - "$tag" <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
d = r["mix_ec_p256"]
print("ec w3 online %.2f passes %s" % (d["online_ms"], [round(x, 2) for x in d.get("passes_online_ms", d.get("passes_total_ms", []))]),
      {k: v for k, v in d["kernel_ms_by_family"].items() if v > 1.0})
f = r.get("operation_length_p256")
if f:
    print("p256 fit e", [round(x, 2) for x in f["executing_ms"]], "v", [round(x, 2) for x in f["verifying_ms"]], "ct/s %.3g" % f["ciphertexts_per_s"][-1])
PY
