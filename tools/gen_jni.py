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

Every Java array / direct buffer is checked against the number of elements the callee reads or writes before the call
(`need_expr` below: the sizes follow from the group's wire widths, the arrays' sizes and the explicit length parameters);
a short one makes the wrapper return VMN_ERR_ARG with a message instead of touching memory beyond it.  Parameters whose
extent the wrapper cannot know are listed in UNCHECKED (tests/test_jni_binding.py pins that list).

No JDK in this image: tests/test_jni_binding.py type-checks the generated C against tests/jni_stub/jni.h (a syntax
stand-in with the JNI specification's signatures) with `gcc -fsyntax-only -Wall -Wextra -Werror`, and checks
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
           "vmn_igen", "vmn_pending")
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


PROOF_TYPES = ("vmn_pos", "vmn_posc", "vmn_ccpos", "vmn_decproof", "vmn_igen")
UNCHECKED = []          # (function, parameter) pairs the wrapper cannot size; filled while generating


def hcast(ptype, pname):
    return f"(({ptype.replace('const ', '').strip()})(intptr_t){pname})"


def group_expr(name, plist):
    """C expression of the vmn_group* the call works in, or None."""
    for ptype, pname in plist:
        t = ptype.replace("const ", "").replace(" const", "").strip()
        if t == "vmn_group*":
            return hcast("vmn_group*", pname)
        if t == "vmn_garray*":
            return f"vmn_garray_group({hcast('vmn_garray*', pname)})"
        if t == "vmn_rarray*":
            return f"vmn_rarray_group({hcast('vmn_rarray*', pname)})"
        for pt in PROOF_TYPES:
            if t == pt + "*":
                return f"{pt}_group({hcast(pt + '*', pname)})"
        if t == "vmn_garray**" and ptype.replace(" ", "").startswith("constvmn_garray*const*"):
            return f"(c_{pname} ? vmn_garray_group((const vmn_garray*)(intptr_t)c_{pname}[0]) : NULL)"
    return None


def first_of(plist, *types):
    for ptype, pname in plist:
        t = ptype.replace("const ", "").replace(" const", "").strip()
        if t in types:
            return t, pname
    return None, None


def need_expr(name, ptype, pname, plist):
    """Number of ELEMENTS (bytes for byte[], ints for int[], ...) the callee touches through this parameter, as a C
    expression over the wrapper's locals; None when it cannot be known here."""
    k = kind(ptype)
    names = [pn for _, pn in plist]
    G = group_expr(name, plist)
    EB, XB = (f"vmnjni_eb({G})", f"vmnjni_xb({G})") if G else (None, None)
    on_rarray = name.startswith("vmn_rarray_")
    at, an = first_of(plist, "vmn_garray*", "vmn_rarray*")
    asize = f"{at[:-1]}_size({hcast(at, an)})" if at else None
    proof = next((pt for pt in PROOF_TYPES if name.startswith(pt + "_")), None)
    pobj = hcast(proof + "*", plist[0][1]) if proof else None
    if k == "handles_out":
        return {"wp_out": "2 * (size_t)width", "factors_out": "2 * (size_t)width", "s_out": "(size_t)width"}.get(pname, "1")
    if k == "handles_in":
        if pname in ("xs", "ys"):
            return "(size_t)k"
        if pname in ("w", "wp", "w_full", "factors"):
            return "2 * (size_t)width"
        if pname in ("s", "s_full"):
            return "(size_t)width"
        if pname == "f":
            return "(size_t)k + 1" if "k" in names else f"(size_t)vmn_decproof_parties({pobj}) + 1"
        if pname == "h" and name == "vmn_igen_set_instance":
            return f"(size_t)vmn_igen_parties({pobj}) + 1"
        return None
    if k in ("ints_in", "ints_out"):
        if pname in ("perm_host",):
            return asize
        if pname == "idx_host":
            return "(size_t)n_out"
        if pname == "idx":
            return "(size_t)n"
        if pname == "layout":
            return "(size_t)items"
        if pname == "negative":
            return "(size_t)threshold"
        if pname == "verdicts5":
            return "5"
        if pname == "pi_out":
            return "(size_t)n"
        if pname == "pi":
            if name == "vmn_permutation_shrink":
                return "(size_t)n_max"
            for cand in ("h", "h_full"):
                if cand in names:
                    return f"vmn_garray_size({hcast('vmn_garray*', cand)})"
            for cand in ("w", "w_full"):
                if cand in names:
                    return f"(c_{cand} ? vmn_garray_size((const vmn_garray*)(intptr_t)c_{cand}[0]) : 0)"
            return None
        return "1"                                  # verdicts, flags, bit counts
    if k in ("longs_in", "longs_out"):
        return "(size_t)items" if pname == "counts" else "1"
    if k == "doubles_out":
        return "1"
    if k in ("bytes_in", "bytes_out", "chars_out"):
        explicit = {"seed": "seedlen", "v_be": "vbytes", "e_be": "ebytes", "rho_be": "rho_bytes", "data": "len", "bt": "len", "buf": "len",
                    "keep": "keep_len"}
        if name == "vmn_keep_list_sanitize" and pname == "keep":
            return "(size_t)(keep_len > n_max ? keep_len : n_max)"       # the trivial list is written over n_max entries
        if pname in explicit and explicit[pname] in names:
            return f"(size_t){explicit[pname]}"
        if pname == "exps_be":
            return f"{asize} * (size_t)ebytes"
        if name == "vmn_prg_bytes" and pname == "out":
            return "(size_t)nbytes"
        if name.startswith("vmn_random_oracle") and pname == "out":
            return "((size_t)nout_bits + 7) / 8"
        if name == "vmn_modp_group_create":
            return "(size_t)nbytes"
        if name == "vmn_group_get_order":
            return XB
        if name == "vmn_group_get_modulus":
            return f"vmnjni_coord({G})"
        if name in ("vmn_msg_push_elements", "vmn_msg_push_ring"):
            return "(size_t)count * (size_t)width"
        if pname == "be":
            return f"(size_t)n * {XB if on_rarray else EB}"
        if pname == "be_out":
            return f"{asize} * {XB if on_rarray else EB}"
        if pname == "out" and name.endswith("_to_bytetree"):
            obj_t, obj_n = plist[0]
            base = obj_t.replace("const ", "").strip()[:-1]
            return f"{base}_bytetree_size({hcast(obj_t, obj_n)})"
        if pname == "e_be" and proof:
            return f"{proof}_size({pobj}) * {XB}"
        if pname in ("keep_host",):
            return asize
        if pname == "keep_out":
            return "(size_t)n_max"
        if pname == "correct":
            return "(size_t)k + 1" if "k" in names else f"(size_t)vmn_decproof_parties({pobj}) + 1"
        if pname == "pkey_be":
            return f"2 * (size_t)width * {EB}"
        if pname == "partials_be":
            return f"(size_t)k * {EB}"
        if pname == "y_be" and name == "vmn_decproof_set_instance":
            return f"((size_t)vmn_decproof_parties({pobj}) + 1) * {EB}"
        if pname == "abs_be":
            return f"(size_t)threshold * {XB}"
        if pname == "out_be" and name == "vmn_rarray_inner_products":
            return f"(size_t)k * vmnjni_xb((c_xs && c_xs[0]) ? vmn_rarray_group((const vmn_rarray*)(intptr_t)c_xs[0]) : NULL)"
        if pname == "out_be" and name == "vmn_pending_finish":
            return f"vmn_pending_bytes({hcast('vmn_pending*', plist[0][1])})"
        if pname == "out_be" and name == "vmn_garray_expprod_multi":
            return f"(size_t)k * {EB}"
        if pname == "out_be" and name == "vmn_pos_get_F":
            return f"2 * vmn_pos_width({pobj}) * {EB}"
        if pname == "out_be" and name == "vmn_ccpos_get_AB":
            return f"(1 + 2 * vmn_ccpos_width({pobj})) * {EB}"
        ring_names = {"last_be", "kx_out", "kx_be", "ka_out", "ka_be", "x_be", "secret_be", "c_be"}
        if pname in ring_names or (on_rarray and pname in ("out_be", "el_be", "v_be")):
            return XB
        elem_names = {"g_be", "base_be", "el_be", "out_be", "a_be", "b_be", "yp_out", "Bp_out", "yp_be", "Bp_be", "Ap_out", "Ap_be",
                      "combinedy_be", "q_be", "p_be"}
        if pname in elem_names and EB:
            return EB
    return None


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
            # the bridge checks every row block the source hands back against n rows of the group's exponent width
            pre.append(f"    vmn_jrs* h_{pname} = {pname} ? vmn_jrs_new(env, {pname}, vmnjni_xb({group_expr(name, plist)})) : NULL;")
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
    body = [f"JNIEXPORT {jret} JNICALL Java_{PKG_JNI}_{cls}_{esc(jname)}({sig}) {{", "    (void)env;", "    (void)cls;"]
    body += pre
    # every array / direct buffer against the elements the callee touches through it
    checks = []
    for ptype, pname in plist:
        k = kind(ptype)
        if k not in ("bytes_in", "bytes_out", "chars_out", "ints_in", "ints_out", "longs_in", "longs_out", "doubles_out", "handles_in",
                     "handles_out"):
            continue
        need = need_expr(name, ptype, pname, plist)
        if need is None:
            if (name, pname) not in UNCHECKED:
                UNCHECKED.append((name, pname))
            continue
        fn = "vmnjni_cap_ok" if pname in direct else "vmnjni_len_ok"
        checks.append(f"{fn}(env, {pname}, {need})")
    callexpr = f"{name}({', '.join(call)})"
    rs_params = [pn for pt, pn in plist if kind(pt) == "rs"]
    if name in RS_OWNERS.values():
        body.append(f"    vmn_jrs_release_owner((void*)(intptr_t){plist[0][1]});      /* the bridge of the random source dies with its object */")
    plain_ret = ret.replace("const ", "").strip()
    if checks:
        body.append("    const int sized = " + "\n                      && ".join(checks) + ";")
        body.append(f'    if (!sized) vmn_report_error("{jname}: a Java array or buffer is shorter than what the call reads or writes");')
    guard = "sized ? " if checks else ""
    if jret == "void":
        body.append(f"    if (sized) {callexpr};" if checks else f"    {callexpr};")
    elif jret_java == "String":
        body.append(f"    const char* res_ = {callexpr};")
    elif plain_ret == "int":
        body.append(f"    {jret} res_ = {guard}({jret}){callexpr}{' : (jint)VMN_ERR_ARG' if checks else ''};")
    elif plain_ret == "size_t":
        body.append(f"    {jret} res_ = {guard}({jret}){callexpr}{' : 0' if checks else ''};")
    else:
        body.append(f"    jlong res_ = {guard}(jlong)(intptr_t){callexpr}{' : 0' if checks else ''};")
    for pn in rs_params:
        if name in RS_OWNERS:
            outp = [p for t, p in plist if kind(t) == "handles_out"][0]
            body.append(f"    if (h_{pn}) {{ if (res_ == 0 && c_{outp}) vmn_jrs_set_owner(h_{pn}, (void*)(intptr_t)c_{outp}[0]); else vmn_jrs_free(env, h_{pn}); }}")
        else:
            body.append(f"    if (h_{pn}) vmn_jrs_free(env, h_{pn});")
    body += post
    if jret_java == "String":
        body.append("    return res_ ? (*env)->NewStringUTF(env, res_) : NULL;")
    elif jret != "void":
        body.append("    return res_;")
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

/* length checks of the wrappers: a null reference is the callee's business (optional parameters; it answers VMN_ERR_ARG) */
static inline int vmnjni_len_ok(JNIEnv* env, jarray a, size_t need) { return !a || (size_t)(*env)->GetArrayLength(env, a) >= need; }
static inline int vmnjni_cap_ok(JNIEnv* env, jobject b, size_t need) {
    if (!b) return 1;
    const jlong cap = (*env)->GetDirectBufferCapacity(env, b);
    return cap >= 0 && (size_t)cap >= need;
}
static inline size_t vmnjni_eb(const vmn_group* g) { return g ? vmn_group_elem_bytes(g) : 0; }
static inline size_t vmnjni_xb(const vmn_group* g) { return g ? vmn_group_exp_bytes(g) : 0; }
static inline size_t vmnjni_coord(const vmn_group* g) { return !g ? 0 : vmn_group_kind(g) == 1 ? vmn_group_elem_bytes(g) / 2 : vmn_group_elem_bytes(g); }
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
    print("unchecked array parameters:", UNCHECKED)


if __name__ == "__main__":
    main()
