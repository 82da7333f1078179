#!/usr/bin/env python3
"""Generate the JNI layer of the two C ABIs: one JNIEXPORT wrapper and one Java `native` declaration per entry point of
include/vmnhip.h and include/vmnproofs.h.

    python3 tools/gen_jni.py          ->  jni/vmnhip_jni.c, jni/vmnproofs_jni.c,
                                          java/com/verificatum/vmnhip/VMNHip.java, VMNProofs.java

The wrappers are mechanical (the C ABI was designed for it: opaque pointers, plain sizes, status codes):

    C parameter                         Java
    vmn_X* / const vmn_X*               long        (opaque handle)
    vmn_X**  (result handles)           long[]      (one slot per result; vmn_shuffle_reencrypt fills 2 * width)
    const vmn_Xarray* const*            long[]
    const uint8_t* / uint8_t*           byte[]      (big-endian rows, byte trees; results are written back)
    const uint32_t* / uint32_t*         int[]       (permutation tables)
    int* / size_t* / long* / double*    int[] / long[] / long[] / double[]   (verdicts, counts, timings)
    const char* / char*                 String / byte[]
    void*                               long        (hipStream_t)
    const vmn_random_source*            RandomSourceBridge  (callbacks into the party's RandomSource)
    const vmn_comm*                     CommBridge          (the all-gather of a sharded proof: byte[] allGather(byte[]))
    int / size_t                        int / long

plus, for the calls that move whole arrays (…_from_be, …_to_be, …_bytetree, vmn_msg_to/from_bytetree), a second
entry point `<name>Direct` taking a direct java.nio.ByteBuffer instead of byte[] (no copy through the Java heap; the
library copies from / to it asynchronously when it is page-locked).

Nothing here can be compiled in this image (no JDK: jni.h is absent); tests/test_jni_binding.py checks
mechanically that every `vmn_*` symbol of the two headers has its wrapper and its native declaration.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "com.verificatum.vmnhip"
PKG_PATH = PKG.replace(".", "/")
PKG_JNI = PKG.replace(".", "_")

HANDLES = ("vmn_ctx", "vmn_group", "vmn_garray", "vmn_rarray", "vmn_msg", "vmn_pos", "vmn_posc", "vmn_ccpos", "vmn_decproof",
           "vmn_igen")
BULK = {"vmn_garray_from_be": ["be"], "vmn_rarray_from_be": ["be"], "vmn_garray_to_be": ["be_out"], "vmn_rarray_to_be": ["be_out"],
        "vmn_garray_to_bytetree": ["out"], "vmn_rarray_to_bytetree": ["out"], "vmn_garray_from_bytetree": ["bt"],
        "vmn_rarray_from_bytetree": ["bt"], "vmn_msg_to_bytetree": ["out"], "vmn_msg_from_bytetree": ["bt"]}
# objects created with a random source keep calling it: the bridge lives until the object's _free
RS_OWNERS = {"vmn_pos_create": "vmn_pos_free", "vmn_posc_create": "vmn_posc_free", "vmn_ccpos_create": "vmn_ccpos_free",
             "vmn_decproof_create": "vmn_decproof_free", "vmn_igen_create": "vmn_igen_free"}
HAND_WRITTEN = {"vmn_msg_item_bytes"}          # const uint8_t** result: returns a byte[] (see the template below)


def prototypes(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", src, flags=re.S)
    src = re.sub(r"typedef enum \w+ \{.*?\} \w+;", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = []
    for m in re.finditer(r"([\w\s\*]+?)\b(vmn_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or "enum" in ret:
            continue
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                mm = re.match(r"(.*?)(\w+)(\[\d*\])?$", p)
                ptype, pname = mm.group(1).strip(), mm.group(2)
                if mm.group(3):
                    ptype += "*"
                plist.append((ptype.replace(" *", "*"), pname))
        out.append((ret.replace(" *", "*"), name, plist))
    return out


def kind(ptype):
    t = ptype.replace("const ", "").replace(" const", "").strip()
    const = "const" in ptype
    if t == "vmn_random_source*":
        return "rs"
    if t == "vmn_comm*":
        return "comm"
    for h in HANDLES:
        if t == h + "*":
            return "handle"
        if t == h + "**":
            return "handles_in" if ptype.replace(" ", "").startswith("const" + h + "*const*") else "handles_out"
    table = {"uint8_t*": "bytes", "uint32_t*": "ints", "int*": "ints", "size_t*": "longs", "long*": "longs", "double*": "doubles",
             "char*": "chars", "void*": "ptr", "int": "int", "size_t": "size", "uint8_t**": "bytes_ptr"}
    if t not in table:
        raise SystemExit(f"gen_jni: unmapped parameter type {ptype!r}")
    k = table[t]
    if k in ("bytes", "ints", "longs", "doubles", "chars"):
        return k + ("_in" if const else "_out")
    return k


JAVA_T = {"handle": "long", "handles_in": "long[]", "handles_out": "long[]", "bytes_in": "byte[]", "bytes_out": "byte[]",
          "ints_in": "int[]", "ints_out": "int[]", "longs_in": "long[]", "longs_out": "long[]", "doubles_out": "double[]",
          "chars_in": "String", "chars_out": "byte[]", "ptr": "long", "int": "int", "size": "long", "rs": "RandomSourceBridge",
          "comm": "CommBridge"}
JNI_T = {"handle": "jlong", "handles_in": "jlongArray", "handles_out": "jlongArray", "bytes_in": "jbyteArray", "bytes_out": "jbyteArray",
         "ints_in": "jintArray", "ints_out": "jintArray", "longs_in": "jlongArray", "longs_out": "jlongArray",
         "doubles_out": "jdoubleArray", "chars_in": "jstring", "chars_out": "jbyteArray", "ptr": "jlong", "int": "jint", "size": "jlong",
         "rs": "jobject", "comm": "jobject"}


def ret_types(ret):
    r = ret.replace("const ", "").strip()
    if r == "int":
        return "int", "jint"
    if r == "size_t":
        return "long", "jlong"
    if r == "void":
        return "void", "void"
    if r == "char*":
        return "String", "jstring"
    return "long", "jlong"                      # pointers: handles, streams


def esc(name):
    return name.replace("_", "_1")


def c_wrapper(cls, ret, name, plist, direct=()):
    """One JNIEXPORT function.  `direct`: names of the byte parameters passed as direct ByteBuffers."""
    jret_java, jret = ret_types(ret)
    jname = name + ("Direct" if direct else "")
    args, pre, call, post = [], [], [], []
    for ptype, pname in plist:
        k = kind(ptype)
        ctype = ptype
        if pname in direct:
            args.append(f"jobject {pname}")
            pre.append(f"    {ctype} c_{pname} = ({ctype})({pname} ? (*env)->GetDirectBufferAddress(env, {pname}) : NULL);")
            call.append(f"c_{pname}")
        elif k == "handle":
            args.append(f"jlong {pname}")
            call.append(f"({ctype})(intptr_t){pname}")
        elif k in ("ptr",):
            args.append(f"jlong {pname}")
            call.append(f"(void*)(intptr_t){pname}")
        elif k == "int":
            args.append(f"jint {pname}")
            call.append(f"(int){pname}")
        elif k == "size":
            args.append(f"jlong {pname}")
            call.append(f"(size_t){pname}")
        elif k == "chars_in":
            args.append(f"jstring {pname}")
            pre.append(f"    const char* c_{pname} = {pname} ? (*env)->GetStringUTFChars(env, {pname}, NULL) : NULL;")
            call.append(f"c_{pname}")
            post.append(f"    if (c_{pname}) (*env)->ReleaseStringUTFChars(env, {pname}, c_{pname});")
        elif k in ("bytes_in", "bytes_out", "chars_out"):
            args.append(f"jbyteArray {pname}")
            pre.append(f"    jbyte* c_{pname} = {pname} ? (*env)->GetByteArrayElements(env, {pname}, NULL) : NULL;")
            call.append(f"({ctype})c_{pname}")
            mode = "JNI_ABORT" if k == "bytes_in" else "0"
            post.append(f"    if (c_{pname}) (*env)->ReleaseByteArrayElements(env, {pname}, c_{pname}, {mode});")
        elif k in ("ints_in", "ints_out"):
            args.append(f"jintArray {pname}")
            pre.append(f"    jint* c_{pname} = {pname} ? (*env)->GetIntArrayElements(env, {pname}, NULL) : NULL;")
            call.append(f"({ctype})c_{pname}")
            mode = "JNI_ABORT" if k == "ints_in" else "0"
            post.append(f"    if (c_{pname}) (*env)->ReleaseIntArrayElements(env, {pname}, c_{pname}, {mode});")
        elif k == "doubles_out":
            args.append(f"jdoubleArray {pname}")
            pre.append(f"    jdouble* c_{pname} = {pname} ? (*env)->GetDoubleArrayElements(env, {pname}, NULL) : NULL;")
            call.append(f"(double*)c_{pname}")
            post.append(f"    if (c_{pname}) (*env)->ReleaseDoubleArrayElements(env, {pname}, c_{pname}, 0);")
        elif k in ("longs_in", "longs_out", "handles_in", "handles_out"):
            # jlong is 64-bit; size_t / long / pointers are 64-bit on every platform this library runs on (checked below)
            args.append(f"jlongArray {pname}")
            pre.append(f"    jlong* c_{pname} = {pname} ? (*env)->GetLongArrayElements(env, {pname}, NULL) : NULL;")
            call.append(f"({ctype})c_{pname}")
            mode = "JNI_ABORT" if k in ("longs_in", "handles_in") else "0"
            post.append(f"    if (c_{pname}) (*env)->ReleaseLongArrayElements(env, {pname}, c_{pname}, {mode});")
        elif k == "rs":
            args.append(f"jobject {pname}")
            pre.append(f"    vmn_jrs* h_{pname} = {pname} ? vmn_jrs_new(env, {pname}) : NULL;")
            pre.append(f"    vmn_random_source s_{pname};")
            pre.append(f"    if (h_{pname}) vmn_jrs_fill(h_{pname}, &s_{pname});")
            call.append(f"h_{pname} ? &s_{pname} : NULL")
        elif k == "comm":
            # the proof object keeps calling the communicator: the bridge is owned by it (first parameter) until its _free
            args.append(f"jobject {pname}")
            pre.append(f"    vmn_jcomm* h_{pname} = {pname} ? vmn_jcomm_new(env, {pname}, (void*)(intptr_t){plist[0][1]}) : NULL;")
            pre.append(f"    vmn_comm s_{pname};")
            pre.append(f"    if (h_{pname}) vmn_jcomm_fill(h_{pname}, &s_{pname});")
            call.append(f"h_{pname} ? &s_{pname} : NULL")
        else:
            raise SystemExit(f"gen_jni: no wrapper rule for {k} ({name}.{pname})")
    sig = ", ".join(["JNIEnv* env", "jclass cls"] + args)
    body = [f"JNIEXPORT {jret} JNICALL Java_{PKG_JNI}_{cls}_{esc(jname)}({sig}) {{", "    (void)cls;"]
    body += pre
    callexpr = f"{name}({', '.join(call)})"
    rs_params = [pn for pt, pn in plist if kind(pt) == "rs"]
    if name in RS_OWNERS.values():
        body.append(f"    vmn_jrs_release_owner((void*)(intptr_t){plist[0][1]});      /* the bridge of the random source dies with its object */")
    if jret == "void":
        body.append(f"    {callexpr};")
    elif jret_java == "String":
        body.append(f"    const char* r = {callexpr};")
    elif ret.replace("const ", "").strip() in ("int", "size_t"):
        body.append(f"    {jret} r = ({jret}){callexpr};")
    else:
        body.append(f"    jlong r = (jlong)(intptr_t){callexpr};")
    for pn in rs_params:
        if name in RS_OWNERS:
            outp = [p for t, p in plist if kind(t) == "handles_out"][0]
            body.append(f"    if (h_{pn}) {{ if (r == 0 && c_{outp}) vmn_jrs_set_owner(h_{pn}, (void*)(intptr_t)c_{outp}[0]); else vmn_jrs_free(env, h_{pn}); }}")
        else:
            body.append(f"    if (h_{pn}) vmn_jrs_free(env, h_{pn});")
    body += post
    if jret_java == "String":
        body.append("    return r ? (*env)->NewStringUTF(env, r) : NULL;")
    elif jret != "void":
        body.append("    return r;")
    body.append("}")
    return "\n".join(body)


def java_decl(ret, name, plist, direct=()):
    jret, _ = ret_types(ret)
    ps = []
    for ptype, pname in plist:
        t = "java.nio.ByteBuffer" if pname in direct else JAVA_T[kind(ptype)]
        ps.append(f"{t} {pname}")
    return f"    public static native {jret} {name}{'Direct' if direct else ''}({', '.join(ps)});"


C_HEAD = '''/* GENERATED by tools/gen_jni.py from include/%(header)s -- do not edit.
 *
 * JNI layer of %(header)s: one wrapper per entry point (see the generator for the type mapping).  Cannot be built in
 * the image this repository is developed in (no JDK); build where one exists:
 *     cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/vmnhip_jni.c jni/vmnproofs_jni.c \\
 *        -Lverificatum-vmn_amd -lvmnproofs -lvmnhip -o libvmnjni.so
 * Status codes are returned as they are; the Java classes turn a negative status into a VMNException carrying
 * vmn_last_error() (INTEGRATION.md).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/%(header)s"
#include "vmnjni_rs.h"

_Static_assert(sizeof(jlong) == sizeof(size_t) && sizeof(jlong) == sizeof(void*) && sizeof(jlong) == sizeof(long),
               "the wrappers pass size_t / long / pointers through jlong");
_Static_assert(sizeof(jint) == sizeof(uint32_t) && sizeof(jint) == sizeof(int), "int[] carries uint32_t tables");
'''

ITEM_BYTES = '''/* vmn_msg_item_bytes: the item's rows as one byte[] (count * width bytes); counts[0] = count, counts[1] = width. */
JNIEXPORT jbyteArray JNICALL Java_%(pkg)s_VMNProofs_vmn_1msg_1item_1bytes(JNIEnv* env, jclass cls, jlong m, jlong i, jlongArray counts) {
    (void)cls;
    const uint8_t* data = NULL;
    size_t count = 0, width = 0;
    if (vmn_msg_item_bytes((const vmn_msg*)(intptr_t)m, (size_t)i, &data, &count, &width) != VMN_OK) return NULL;
    jbyteArray out = (*env)->NewByteArray(env, (jsize)(count * width));
    if (!out) return NULL;
    (*env)->SetByteArrayRegion(env, out, 0, (jsize)(count * width), (const jbyte*)data);
    if (counts) {
        jlong cw[2] = {(jlong)count, (jlong)width};
        (*env)->SetLongArrayRegion(env, counts, 0, 2, cw);
    }
    return out;
}
'''


def emit(header, cls):
    protos = prototypes(header)
    c = [C_HEAD % {"header": header}]
    j = [f"// GENERATED by tools/gen_jni.py from include/{header} -- do not edit.",
         f"package {PKG};", "",
         f"/** The entry points of include/{header}, one {'{@code native}'} method each (same names, status codes returned as they are).",
         " *  Handles are {@code long}; result handles come back through {@code long[]} slots.  See tools/gen_jni.py for the",
         " *  type mapping and INTEGRATION.md for the classes built on top. */",
         f"public final class {cls} {{", f"    private {cls}() {{ }}", "",
         "    static {", '        System.loadLibrary("vmnjni");', "    }", ""]
    for ret, name, plist in protos:
        if name in HAND_WRITTEN:
            c.append(ITEM_BYTES % {"pkg": PKG_JNI})
            j.append("    public static native byte[] vmn_msg_item_bytes(long m, long i, long[] counts);")
            continue
        c.append(c_wrapper(cls, ret, name, plist))
        c.append("")
        j.append(java_decl(ret, name, plist))
        if name in BULK:
            c.append(c_wrapper(cls, ret, name, plist, direct=BULK[name]))
            c.append("")
            j.append(java_decl(ret, name, plist, direct=BULK[name]))
    j.append("}")
    return "\n".join(c) + "\n", "\n".join(j) + "\n", [p[1] for p in protos]


def main():
    os.makedirs(os.path.join(ROOT, "jni"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "java", PKG_PATH), exist_ok=True)
    for header, cls, cfile in (("vmnhip.h", "VMNHip", "vmnhip_jni.c"), ("vmnproofs.h", "VMNProofs", "vmnproofs_jni.c")):
        c, j, names = emit(header, cls)
        open(os.path.join(ROOT, "jni", cfile), "w").write(c)
        open(os.path.join(ROOT, "java", PKG_PATH, cls + ".java"), "w").write(j)
        print(f"{header}: {len(names)} entry points -> jni/{cfile}, java/{PKG_PATH}/{cls}.java")


if __name__ == "__main__":
    main()
