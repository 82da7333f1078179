#!/bin/bash
# Registers, scratch and occupancy of every kernel as the compiler reports them (-Rpass-analysis=kernel-resource-usage);
# prints the kernels that spill (ScratchSize > 0) or run below two waves per SIMD.  CPU only (cross-compiles gfx950).
# usage: tools/resource_usage.sh [unit ...]        default: all units
cd "$(dirname "$0")/.." || exit 1
out=${TMPDIR:-/tmp}/vmn_resource_usage; mkdir -p "$out"
units=${*:-vmnhip inst_small inst_2048 inst_2048_wide inst_3072 inst_4096 inst_8192 inst_16384 inst_p224 inst_p256 inst_p384 inst_p521}
for u in $units; do
  /opt/rocm/bin/hipcc -std=c++20 -O3 --offload-arch=gfx950 -fPIC -c verificatum-vmn_amd/csrc/$u.hip -o "$out/$u.o" \
      -Rpass-analysis=kernel-resource-usage 2> "$out/$u.log" &
done
wait
python3 - "$out" <<'PY'
import glob, re, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    for b in re.split(r"remark: [^\n]*Function Name: ", open(f).read())[1:]:
        name = b.split()[0]
        get = lambda k: (lambda m: int(m.group(1)) if m else None)(re.search(k + r": (\d+)", b))
        sc, occ = get(r"ScratchSize \[bytes/lane\]"), get(r"Occupancy \[waves/SIMD\]")
        if sc or (occ is not None and occ < 2):
            print(f"{f.split('/')[-1][:-4]:16s} {name[:80]:80s} VGPR {get('VGPRs')} AGPR {get('AGPRs')} scratch {sc} B/lane occupancy {occ}")
PY
