#!/usr/bin/env python3
"""Copy the data fixtures the reference tree holds for this path into tests/golden/ (data, not code; run in the build
container -- the reference is not present on the GPU box):

  * /root/reference/demo/mixnet/benchmarks/bench_config:43 contains (commented out) a marshalled 15 492-bit safe-prime
    ModPGroup as a hex byte tree.  It pins the byte-tree wire format and the fixed-width integer
    encoding that the import/export kernels use.           -> reference_modpgroup_bytetree.hex
  * /root/reference/demo/mixnet/group_descriptions:29-32 defines two groups by explicit parameters (`vog -gen ModPGroup
    -explic <modulus> <generator> [<order>]`): ModPGroup_1024_256 -- a 1024-bit p with a 256-bit prime-order subgroup, the
    one ModPGroup of the tree that is NOT a safe-prime group (q != (p - 1) / 2: membership is x^q = 1, generators are
    t^((p-1)/q)) -- and ModPGroup_safeprime_15492, the same 15 492-bit group as the byte tree above, as p and g in hex:
    a second, independent statement of the fixture.          -> reference_group_descriptions.json"""
import json
import os
import re

SRC = "/root/reference/demo/mixnet/benchmarks/bench_config"
GROUPS = "/root/reference/demo/mixnet/group_descriptions"
HERE = os.path.dirname(os.path.abspath(__file__))


def bytetree_fixture():
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


def group_descriptions():
    src = open(GROUPS).read()
    out = {"source": "demo/mixnet/group_descriptions:29-32 (vog -gen ModPGroup -explic <modulus> <generator> [<order>])", "groups": {}}
    for m in re.finditer(r"^(\w+)=\$\(vog -gen ModPGroup (.*?)\)\s*$", src, flags=re.M | re.S):
        hexes = re.findall(r'"([0-9a-f]+)"', m.group(2))
        rec = {"p": hexes[0], "g": hexes[1], "vog_flags": re.findall(r"-\w+", m.group(2))}
        if len(hexes) > 2:
            rec["q"] = hexes[2]
        out["groups"][m.group(1)] = rec
    path = os.path.join(HERE, "reference_group_descriptions.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(path, sorted(out["groups"]))


def main():
    bytetree_fixture()
    group_descriptions()


if __name__ == "__main__":
    main()
