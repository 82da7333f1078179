"""CPU suite: the JNI / Java binding is complete source, checked mechanically -- there is no JDK in
this image, so nothing is linked or run; the C is type-checked against tests/jni_stub/jni.h (a SYNTAX stand-in, see its
header), and:

  * every `vmn_*` entry point of include/vmnhip.h and include/vmnproofs.h has a JNIEXPORT wrapper in jni/*.c and a
    `native` declaration in the matching Java class, with the same number of parameters;
  * the generated files are exactly what tools/gen_jni.py produces from the headers today (no drift);
  * every native the hand-written Java classes call exists, with the right number of arguments;
  * the classes of seam S1 exist and implement the reference's interfaces with the reference's method signatures
    (src/java/com/verificatum/protocol/hvzk/{PoS,PoSC,CCPoS}.java and their factories)."""
import importlib.util
import os
import re

from conftest import ROOT
from test_abi_exports import declared_symbols

JAVA = os.path.join(ROOT, "java", "com", "verificatum", "vmnhip")
PAIRS = (("vmnhip.h", "VMNHip", "vmnhip_jni.c"), ("vmnproofs.h", "VMNProofs", "vmnproofs_jni.c"))


def esc(name):
    return name.replace("_", "_1")


def gen():
    spec = importlib.util.spec_from_file_location("gen_jni", os.path.join(ROOT, "tools", "gen_jni.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def natives(cls):
    """{name: parameter count} of the native methods of a generated class."""
    text = open(os.path.join(JAVA, cls + ".java")).read()
    out = {}
    for m in re.finditer(r"public static native [\w\[\]\.]+ (\w+)\(([^)]*)\);", text):
        out[m.group(1)] = len([p for p in m.group(2).split(",") if p.strip()])
    return out


def split_args(s):
    """Top-level comma split of a Java argument list."""
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return args


def test_every_entry_point_has_a_wrapper_and_a_native_declaration():
    g = gen()
    for header, cls, cfile in PAIRS:
        csrc = open(os.path.join(ROOT, "jni", cfile)).read()
        nat = natives(cls)
        protos = {name: plist for _, name, plist in g.prototypes(header)}
        syms = declared_symbols(header)
        assert set(protos) == set(syms), (set(syms) ^ set(protos))      # the generator's parser sees what the export test sees
        for name in syms:
            m = re.search(r"JNIEXPORT \w+ JNICALL Java_com_verificatum_vmnhip_%s_%s\(([^)]*)\)" % (cls, esc(name)), csrc)
            assert m, f"{name}: no JNIEXPORT wrapper in jni/{cfile}"
            assert name in nat, f"{name}: no native declaration in {cls}.java"
            if name != "vmn_msg_item_bytes":                             # hand-written: returns the rows as a byte[]
                jni_params = len(split_args(m.group(1))) - 2             # JNIEnv*, jclass
                assert jni_params == len(protos[name]) == nat[name], name
                assert re.search(r"\b%s\(" % name, csrc[m.end():m.end() + 4000]), f"{name}: the wrapper does not call it"


def test_generated_files_are_current():
    g = gen()
    for header, cls, cfile in PAIRS:
        c, j, _ = g.emit(header, cls)
        assert open(os.path.join(ROOT, "jni", cfile)).read() == c, f"jni/{cfile} is stale: run tools/gen_jni.py"
        assert open(os.path.join(JAVA, cls + ".java")).read() == j, f"{cls}.java is stale: run tools/gen_jni.py"


def test_hand_written_classes_call_existing_natives_with_the_right_arity():
    nat = {"VMNHip": natives("VMNHip"), "VMNProofs": natives("VMNProofs")}
    calls = 0
    for fname in sorted(os.listdir(JAVA)):
        if fname in ("VMNHip.java", "VMNProofs.java"):
            continue
        text = open(os.path.join(JAVA, fname)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//.*", "", text)
        for m in re.finditer(r"\b(VMNHip|VMNProofs)\.(vmn_\w+)\(", text):
            cls, name = m.group(1), m.group(2)
            assert name in nat[cls], f"{fname}: {cls}.{name} is not a native of {cls}"
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            assert len(split_args(text[m.end():i - 1])) == nat[cls][name], f"{fname}: {cls}.{name} called with the wrong number of arguments"
            calls += 1
    assert calls > 80


def test_seam_classes_mirror_the_reference_interfaces():
    want = {"PoSGPU.java": ["implements PoS", "void precompute(final Log log, final PGroupElement g, final PGroupElementArray h, final Permutation pi)",
                            "void prove(final Log log, final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp",
                            "void precompute(final Log log, final PGroupElement g, final PGroupElementArray h)",
                            "boolean verify(final Log log, final int l, final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp)",
                            "void free()", '"PermutationCommitment"', '"Commitment"', '"Reply"', "PoSCommitment", "PoSReply"],
            "PoSCGPU.java": ["implements PoSC", "void prove(final Log log, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u",
                             "boolean verify(final Log log, final int l, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u)",
                             "PoSCCommitment", "PoSCReply"],
            "CCPoSGPU.java": ["implements CCPoS", "final PGroupElementArray raisedu, final PGroupElementArray raisedh, final PRingElement raisedExponent",
                              "CCPoSCommitment", "CCPoSReply", "vmn_ctx_helper_begin"],
            "PoSGPUFactory.java": ["implements PoSFactory", "PoS newPoS(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp)"],
            "PoSCGPUFactory.java": ["implements PoSCFactory", "PoSC newPoSC("],
            "CCPoSGPUFactory.java": ["implements CCPoSFactory", "CCPoS newCCPoS("],
            "PGroupElementArrayGPU.java": ["exp(", "expProd(", "mul(", "prod()", "permute(", "shiftPush(", "copyOfRange(", "extract(", "free()"],
            "RandomSourceBridge.java": ["byte[] ringElements(long n)", "byte[] integers(long n, int bits)", "byte[] arraySeed()", "boolean deviceArrays()"]}
    for fname, needles in want.items():
        text = " ".join(open(os.path.join(JAVA, fname)).read().split())
        for n in needles:
            assert " ".join(n.split()) in text, f"{fname}: missing {n!r}"
    rs = open(os.path.join(ROOT, "jni", "vmnjni_rs.c")).read()
    for method, sig in (("ringElements", "(J)[B"), ("integers", "(JI)[B"), ("arraySeed", "()[B"), ("deviceArrays", "()Z")):
        assert f'"{method}", "{sig}"' in rs                              # the bridge looks up exactly the interface's methods


def test_jni_sources_pass_a_compiler():
    """gcc -fsyntax-only -Wall -Wextra -Werror over jni/*.c with the test-only jni.h stand-in: a C error anywhere in the
    generated wrappers or the hand-written bridges fails the CPU suite."""
    import subprocess
    stub = os.path.join(ROOT, "tests", "jni_stub")
    assert "SYNTAX STAND-IN" in open(os.path.join(stub, "jni.h")).read()
    srcs = [os.path.join(ROOT, "jni", f) for f in ("vmnjni_rs.c", "vmnhip_jni.c", "vmnproofs_jni.c")]
    pr = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-comment", "-I" + stub,
                         "-I" + os.path.join(ROOT, "include")] + srcs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert pr.returncode == 0, pr.stdout.decode()[-4000:]
    # the stand-in is test infrastructure: nothing shipped includes it
    for d in ("jni", "include", "verificatum-vmn_amd"):
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".c", ".h", ".cpp", ".hip", ".py")):
                    text = open(os.path.join(base, f), errors="replace").read()
                    assert not re.search(r"#\s*include[^\n]*jni_stub", text) and "-Itests/jni_stub" not in text, f


def test_every_java_array_is_checked_against_what_the_callee_touches():
    """vmn_garray_to_be writes n * nbytes into be_out whatever its length.  The generator sizes every array
    and direct-buffer parameter (none is left unchecked) and the wrappers refuse a short one."""
    g = gen()
    for header, cls, cfile in PAIRS:
        g.emit(header, cls)
    assert g.UNCHECKED == []
    csrc = open(os.path.join(ROOT, "jni", "vmnhip_jni.c")).read()
    body = csrc[csrc.index("VMNHip_vmn_1garray_1to_1be("):]
    body = body[:body.index("\n}\n")]
    assert "vmnjni_len_ok(env, be_out, vmn_garray_size(" in body and "vmnjni_eb(vmn_garray_group(" in body and "VMN_ERR_ARG" in body
    direct = csrc[csrc.index("VMNHip_vmn_1garray_1to_1beDirect("):]
    assert "vmnjni_cap_ok(env, be_out," in direct[:direct.index("\n}\n")]
    n_checks = sum(open(os.path.join(ROOT, "jni", f)).read().count("const int sized =") for f in ("vmnhip_jni.c", "vmnproofs_jni.c"))
    assert n_checks > 120
    rs = open(os.path.join(ROOT, "jni", "vmnjni_rs.c")).read()
    assert "GetArrayLength(env, arr) < n * h->row_bytes" in rs
