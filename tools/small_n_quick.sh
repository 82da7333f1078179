#!/bin/bash
# The small sizes of bench.py alone: PoS over the 2048-bit group at N = 10^4 (BASELINE configs[0]) and the P-256 fit up to 10^4.
# usage (GPU box): bash tools/small_n_quick.sh <tag>
tag=${1:-small_n}
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements 10000 --ec-elements 10000 --ccpos-elements 0 \
    --decrypt-elements 0 --no-e2e --skip-cpu --no-shapes > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -5 gpurun_out/$tag.err; exit 1; }
python - "$tag" <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
d = r["mix_prove"]
print("PoS-2048 N=10^4: %.2f ms passes %s (re-encrypt %.2f prove %.2f verify %.2f) launches %s" % (d["total_ms"], d["passes_total_ms"], d["reencrypt_ms"], d["prove_ms"], d["verify_ms"], d.get("kernel_launches")))
f = r.get("operation_length_p256")
if f:
    print("p256 fit e", [round(x, 2) for x in f["executing_ms"]], "v", [round(x, 2) for x in f["verifying_ms"]])
e = r.get("mix_ec_p256")
if e:
    print("ec w3 N=10^4 online %.2f" % e["online_ms"], e.get("passes_online_ms"))
PY
