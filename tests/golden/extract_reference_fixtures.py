#!/usr/bin/env python3
"""Copy the one data fixture the reference tree holds for this path into tests/golden/.

/root/reference/demo/mixnet/benchmarks/bench_config:43 contains (commented out) a marshalled
15 492-bit safe-prime ModPGroup as a hex byte tree: data, not code.  It pins the byte-tree wire
format and the fixed-width integer encoding that the import/export kernels use.
Run in the build container (the reference is not present on the GPU box)."""
import os
import re

SRC = "/root/reference/demo/mixnet/benchmarks/bench_config"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for line in open(SRC):
        m = re.search(r"define\(BENCH_PGROUP,\s*([0-9a-f]{1000,})", line)
        if m:
            hexstr = m.group(1)
            out = os.path.join(HERE, "reference_modpgroup_bytetree.hex")
            with open(out, "w") as f:
                f.write(hexstr + "\n")
            print(out, len(hexstr) // 2, "bytes")
            return
    raise SystemExit("fixture not found")


if __name__ == "__main__":
    main()
