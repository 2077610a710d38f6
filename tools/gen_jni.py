#!/usr/bin/env python3
"""Generates the JNI shim and the Scala `Native` object from include/dismember_hip.h, one native method per C entry point.

    python tools/gen_jni.py            # rewrites jni/dismember_jni.c and scala/com/mass/hip/Native.scala
    python tools/gen_jni.py --check    # exit 1 if the committed files differ from what the header yields

tests/test_jni_shim.py runs the check (so the shim cannot drift from the header) and cross-checks the method set against
dismember_amd/_native.SIGNATURES.  The reference reaches native code the same way (BigDL's MKL JNI,
project/Dependencies.scala:27-29); nothing here is compiled in the build image (no JVM), the shim is compile-guarded on
JAVA_HOME by jni/Makefile.

Mapping: handle / communicator / device pointer -> jlong; `const T *` host arrays -> primitive arrays (released with
JNI_ABORT when const, copied back otherwise; null allowed); `T *out` scalars -> arrays of length 1; option structs -> their
fields as scalars; host `void *` buffers -> one method per element type; a non-zero status becomes the exception the Scala
code would have thrown (raise(); communicator calls report dm_comm_last_error).

Array access: the JNI spec forbids blocking (on other threads, the network, or a long-running device) inside a
Get/ReleasePrimitiveArrayCritical region — a thread parked in a collective while another is stalled by the GC locker
deadlocks the JVM.  Only the entry points listed in CRITICAL_OK (pure host logic and plain copies) pin their arrays that
way; every other call goes through Get<Type>ArrayElements / Release<Type>ArrayElements, which may copy.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "dismember_hip.h")
OUT_C = os.path.join(ROOT, "jni", "dismember_jni.c")
OUT_SCALA = os.path.join(ROOT, "scala", "com", "mass", "hip", "Native.scala")

STRUCTS = {      # option structs flattened into scalars: (field, C type) in declaration order
    "dm_tdm_search_opts": [("beam", "int"), ("topk", "int"), ("use_mask", "int"), ("widen_consumed", "int")],
    "dm_adam_opts": [("lr", "double"), ("lr_decay", "double"), ("beta1", "double"), ("beta2", "double"), ("eps", "double")],
    "dm_otm_train_opts": [("beam", "int"), ("leaf_level", "int"), ("use_mask", "int"), ("target_mode", "int")],
    "dm_sample_opts": [("start_level", "int"), ("with_prob", "int"), ("tolerance", "int"), ("use_mask", "int"), ("seed", "uint64_t")],
}
SCALAR = {"int": ("jint", "Int"), "int32_t": ("jint", "Int"), "int64_t": ("jlong", "Long"), "uint64_t": ("jlong", "Long"),
          "size_t": ("jlong", "Long"), "float": ("jfloat", "Float"), "double": ("jdouble", "Double")}
ARRAY = {"int32_t": ("jintArray", "jint", "Array[Int]"), "int": ("jintArray", "jint", "Array[Int]"),
         "uint32_t": ("jintArray", "jint", "Array[Int]"), "int64_t": ("jlongArray", "jlong", "Array[Long]"),
         "uint64_t": ("jlongArray", "jlong", "Array[Long]"), "float": ("jfloatArray", "jfloat", "Array[Float]"),
         "double": ("jdoubleArray", "jdouble", "Array[Double]"), "uint8_t": ("jbyteArray", "jbyte", "Array[Byte]")}
# host `void *` arguments: entry point -> {arg: [(method suffix, element C type, extra fixed call args)]}
VOID_HOST = {
    "dm_load_weights_din": {"compact": [("F32", "float"), ("F64", "double")]},     # dtype is fixed by the variant (FIXED below)
    "dm_din_forward": {"logits": [("F32", "float"), ("F64", "double")]},
    "dm_train_download": {"out": [("F32", "float"), ("F64", "double")]},               # the loaded dtype
    "dm_comm_unique_id": {"id128": [("", "uint8_t")]},
    "dm_comm_create_rccl": {"id128": [("", "uint8_t")]},
    "dm_comm_all_gather_v": {"send": [("", "uint8_t")], "recv": [("", "uint8_t")]},
    # typed variants: a Scala Array[Int] / Array[Float] goes up (comes down) as it is — no ByteBuffer copy on the caller's side
    "dm_memcpy_h2d": {"src": [("", "uint8_t"), ("I32", "int32_t"), ("F32", "float"), ("F64", "double")]},
    "dm_memcpy_d2h": {"dst": [("", "uint8_t"), ("I32", "int32_t"), ("F32", "float"), ("F64", "double")]},
}
FIXED = {("dm_load_weights_din", "F32"): {"dtype": "DM_F32"}, ("dm_load_weights_din", "F64"): {"dtype": "DM_F64"}}   # args the variant pins
HANDWRITTEN = {"dm_dr_load_model"}          # struct with pointer arrays: written out below
# entry points that neither wait on peers / the network nor run a long device job: the only ones allowed a Critical region
# (not dm_create — HIP runtime and device initialisation can take seconds — and not dm_memcpy_h2d / _d2h: arbitrarily large
# synchronous copies; both would hold the GC locker for their whole duration)
CRITICAL_OK = {"dm_level_start", "dm_jtm_shard_range", "dm_tdm_id_to_code", "dm_kernel_timing_get",
               "dm_kernel_timing_get_kind", "dm_get_scorer_mode", "dm_comm_rank", "dm_device_count", "dm_last_scored_rows",
               "dm_train_last_loss", "dm_train_sync_stats", "dm_jtm_last_step_seconds", "dm_adam_last_step_rows", "dm_comm_unique_id", "dm_dev_alloc"}
JTYPE = {"jint": "Int", "jlong": "Long", "jfloat": "Float", "jdouble": "Double", "jbyte": "Byte"}
# Minimum lengths (in elements) of host arrays whose extent follows from the scalar arguments of the same call: checked with
# GetArrayLength BEFORE anything is pinned, so that a caller mistake raises IllegalArgumentException instead of overrunning the
# JVM heap (the C side trusts its (pointer, size) pairs).  Expressions are over the C argument names (opts_* = struct fields);
# arrays whose extent depends on the CONTENTS of another array (CSR targets, consumed ids) are checked by the library where it can.
_SEARCH_OUT = {"out_item_ids": "U * opts_topk", "out_scores": "U * opts_topk", "out_counts": "U"}
_OTM_OUT = {"seq_codes": "U * L", "out_node_ids": "U * 2 * beam", "out_scores": "U * 2 * beam", "out_counts": "U"}
EXTENTS = {
    "dm_load_tree_tdm": {"codes": "n_nodes", "node_ids": "n_nodes", "is_leaf": "n_nodes"},
    "dm_load_id_maps": {"leaf_item_ids": "n", "leaf_codes": "n"},
    "dm_tdm_id_to_code": {"item_ids": "n", "codes": "n", "mask_pos": "n", "n_mask": "1"},
    "dm_din_forward": {"codes": "B", "seqs": "B * L", "pad_flat_idx": "n_pad", "logits": "B"},
    "dm_tdm_beam_search": dict(_SEARCH_OUT, seq_item_ids="U * L", consumed_off="U + 1"),
    "dm_tdm_beam_search_trace": dict(_SEARCH_OUT, seq_item_ids="U * L", trace_counts="U * max_levels"),
    "dm_otm_beam_search": _OTM_OUT, "dm_otm_beam_search_f64": _OTM_OUT,
    "dm_otm_beam_search_trace": dict(_OTM_OUT, trace_counts="U * max_levels"),
    "dm_otm_beam_search_trace_f64": dict(_OTM_OUT, trace_counts="U * max_levels"),
    "dm_tdm_bruteforce_topk": {"seq_item_ids": "U * L", "out_item_ids": "U * topk", "out_scores": "U * topk", "out_counts": "U"},
    "dm_jtm_child_weights": {"row_off": "n_items + 1", "item_node": "n_items", "weights": "n_items * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))"},
    "dm_jtm_cache_rows": {"row_off": "n_items + 1"},
    "dm_jtm_cache_rows_range": {"row_off": "n_items + 1"},
    "dm_jtm_shard_range": {"i_lo": "1", "i_hi": "1"},
    "dm_jtm_child_weights_cached": {"item_node": "n_items", "weights": "n_items * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))"},
    "dm_jtm_step_cached": {"item_node": "n_items", "old_node": "n_items", "out_node": "n_items"},
    "dm_jtm_optimize_cached": {"item_code": "n_items", "out_proj": "n_items"},
    "dm_jtm_optimize_all": {"hs": "n", "item_code": "n_items", "out_proj": "n_items"},
    "dm_jtm_rebalance": {"weights": "n * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))", "old_node": "n", "out_node": "n"},
    "dm_jtm_rebalance_all": {"weights": "n * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))", "old_node": "n", "item_node": "n", "out_node": "n"},
    "dm_otm_rebalance_all": {"weights": "n * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))", "old_node": "n", "item_node": "n", "out_node": "n"},
    "dm_otm_child_weights": {"row_off": "n_items + 1", "item_node": "n_items", "weights": "n_items * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))"},
    "dm_otm_rebalance": {"weights": "n * ((jlong)1 << (level > old_level && level - old_level < 32 ? level - old_level : 0))", "old_node": "n", "out_node": "n"},
    "dm_train_forward_backward": {"codes": "B", "seqs": "B * L", "pad_flat_idx": "n_pad", "labels": "B", "loss": "1"},
    "dm_comm_create_all": {"devices": "n", "out": "n"},
    "dm_comm_allreduce_f64": {"vals": "n"},
    "dm_otm_train_batch": {"seq_codes": "U * L", "target_off": "U + 1"},
    "dm_otm_pseudo_targets": {"seq_codes": "U * L", "target_off": "U + 1", "out_counts": "U"},
    "dm_tdm_set_node_probs": {"codes": "n", "probs": "n"},
    "dm_tdm_make_train_batch": {"seq_item_ids": "T * L", "target_item_ids": "T", "neg_counts": "n_counts", "out_codes": "cap",
                                "out_seqs": "cap * L", "out_rowmask": "cap", "out_labels": "cap", "n_rows": "1"},
    "dm_tdm_sample_train_batch_dev": {"neg_counts": "n_counts", "n_rows": "1"},
    "dm_dr_load_path_items": {"item_off": "n_paths + 1"},
    "dm_allreduce_grads": {"hs": "n"},
}


def pin(cname, je, an, const):
    """(acquire, release) statements for the host array `an` of JNI element type `je`."""
    mode = "JNI_ABORT" if const else "0"
    if cname in CRITICAL_OK:
        return ("  %s *p_%s = %s ? (*e)->GetPrimitiveArrayCritical(e, %s, 0) : 0;" % (je, an, an, an),
                "  if (p_%s) (*e)->ReleasePrimitiveArrayCritical(e, %s, p_%s, %s);" % (an, an, an, mode))
    t = JTYPE[je]
    return ("  %s *p_%s = %s ? (*e)->Get%sArrayElements(e, %s, 0) : 0;" % (je, an, an, t, an),
            "  if (p_%s) (*e)->Release%sArrayElements(e, %s, p_%s, %s);" % (an, t, an, an, mode))


def camel(name):
    parts = name[3:].split("_")
    return parts[0] + "".join(p.capitalize() for p in parts[1:])


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = []
    for m in re.finditer(r"\b(int|const char \*)\s*(dm_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        alist = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(const\s+)?([\w ]+?)\s*(\*\s*(?:const\s*)?\*?)?\s*(\w+)$", a)
                const, base, ptr, an = bool(mm.group(1)), mm.group(2).strip(), (mm.group(3) or "").replace(" ", ""), mm.group(4)
                alist.append(dict(const=const, base=base, ptr=ptr, name=an))
        protos.append((ret, name, alist))
    return protos


def expand(protos):
    """One or more (method name, C name, ret, args, void element type per arg) per prototype."""
    out = []
    for ret, name, args in protos:
        if name in HANDWRITTEN:
            continue
        variants = [("", {})]
        for an, vs in VOID_HOST.get(name, {}).items():
            variants = [(s + suf, dict(d, **{an: ct})) for s, d in variants for suf, ct in vs]
        # keep suffixes unique when two void args share the same element type
        seen = {}
        for suf, d in variants:
            seen.setdefault(suf, d)
        for suf, d in seen.items():
            out.append((camel(name) + suf, name, ret, args, dict(d, __fixed__=FIXED.get((name, suf), {}))))
    return out


def gen(protos):
    c, sc = [], []
    for meth, cname, ret, args, voids in expand(protos):
        jparams, sparams, pre, post, call = [], [], [], [], []
        pinned = []            # host arrays acquired from the JVM: a NULL for a non-null array means a pending OutOfMemoryError
        handle_expr, comm_expr = "0", None
        for i, a in enumerate(args):
            base, ptr, an, const = a["base"], a["ptr"], a["name"], a["const"]
            if an in voids.get("__fixed__", {}):
                call.append(voids["__fixed__"][an])
            elif base in ("dm_handle_t", "dm_comm_t") and ptr == "":
                jparams.append("jlong %s" % an); sparams.append("%s: Long" % an)
                call.append("(%s)(intptr_t)%s" % (base, an))
                if base == "dm_handle_t" and i == 0:
                    handle_expr = "(dm_handle_t)(intptr_t)%s" % an
                if base == "dm_comm_t" and i == 0:
                    comm_expr = "(dm_comm_t)(intptr_t)%s" % an
            elif base in ("dm_handle_t", "dm_comm_t") and ptr == "*":      # out handle(s) or a list of handles
                jparams.append("jlongArray %s" % an); sparams.append("%s: Array[Long]" % an)
                a_, r_ = pin(cname, "jlong", an, False)
                pre.append(a_); post.append(r_); pinned.append(an)
                call.append("(%s *)p_%s" % (base, an))
            elif base in STRUCTS and ptr == "*":
                fields = STRUCTS[base]
                for f, ct in fields:
                    jparams.append("%s %s_%s" % (SCALAR[ct][0], an, f)); sparams.append("%s%s: %s" % (an, f.title().replace("_", ""), SCALAR[ct][1]))
                pre.append("  %s s_%s = { %s };" % (base, an, ", ".join("(%s)%s_%s" % (ct, an, f) for f, ct in fields)))
                call.append("&s_%s" % an)
            elif base == "char" and ptr == "*":
                jparams.append("jstring %s" % an); sparams.append("%s: String" % an)
                pre.append("  const char *p_%s = %s ? (*e)->GetStringUTFChars(e, %s, 0) : 0;" % (an, an, an))
                post.append("  if (p_%s) (*e)->ReleaseStringUTFChars(e, %s, p_%s);" % (an, an, an)); pinned.append(an)
                call.append("p_%s" % an)
            elif base == "void" and ptr == "*" and an in voids:               # host buffer of a known element type
                jt, je, st = ARRAY[voids[an]]
                jparams.append("%s %s" % (jt, an)); sparams.append("%s: %s" % (an, st))
                a_, r_ = pin(cname, je, an, const)
                pre.append(a_); post.append(r_); pinned.append(an)
                call.append("p_%s" % an)
            elif ptr in ("*", "**") and (base == "void" or an.startswith("d_") or an == "dptr"):   # device pointers travel as jlong
                if ptr == "**" or (base != "void" and an in ("d_ptr",)) or an == "dptr":
                    jparams.append("jlongArray %s" % an); sparams.append("%s: Array[Long]" % an)
                    a_, r_ = pin(cname, "jlong", an, False)
                    pre.append(a_); post.append(r_); pinned.append(an)
                    call.append("(%s %s)p_%s" % (base, ptr, an))
                else:
                    jparams.append("jlong %s" % an); sparams.append("%s: Long" % an)
                    call.append("(%s%s *)(intptr_t)%s" % ("const " if const else "", base, an))
            elif ptr == "*" and base in ARRAY:
                jt, je, st = ARRAY[base]
                jparams.append("%s %s" % (jt, an)); sparams.append("%s: %s" % (an, st))
                a_, r_ = pin(cname, je, an, const)
                pre.append(a_); post.append(r_); pinned.append(an)
                call.append("(%s%s *)p_%s" % ("const " if const else "", base, an))
            elif ptr == "" and base in SCALAR:
                jparams.append("%s %s" % (SCALAR[base][0], an)); sparams.append("%s: %s" % (an, SCALAR[base][1]))
                call.append("(%s)%s" % (base, an))
            else:
                raise SystemExit("gen_jni: cannot map argument `%s %s%s` of %s" % (base, ptr, an, cname))
        jret, sret = ("jstring", "String") if ret != "int" else (("jint", "Int") if cname in ("dm_version",) else ("void", "Unit"))
        sig = "JNIEXPORT %s JNICALL Java_com_mass_hip_Native_%s(JNIEnv *e, jclass cls%s) {" % (jret, meth.replace("_", "_1"), "".join(", " + p for p in jparams))
        checks = []
        names = set(a["name"] for a in args)
        for an, expr in EXTENTS.get(cname, {}).items():
            if an not in names:
                raise SystemExit("gen_jni: EXTENTS names `%s`, which %s does not take" % (an, cname))
            need = re.sub(r"\bopts_(\w+)", r"opts_\1", expr)
            checks.append('  if (%s && (jlong)(*e)->GetArrayLength(e, %s) < (jlong)(%s)) { raise_msg(e, DM_ERR_INVALID, "%s: array `%s` is shorter than %s"); (void)cls; return; }'
                          % (an, an, need, meth, an, expr.replace("(jlong)", "")))
        body = [sig] + checks + pre
        if pinned:
            # the JVM could not hand out one of the arrays (an exception is already pending): give back what was acquired and
            # return to Java without calling into the library or throwing on top of it
            body.append("  if (%s) {" % " || ".join("(%s && !p_%s)" % (an, an) for an in pinned))
            body += ["  " + ln.replace(", 0);", ", JNI_ABORT);") for ln in post[::-1]]      # nothing was written: no copy-back
            body.append("    (void)cls; return%s;" % (" 0" if ret != "int" or cname in ("dm_version",) else ""))
            body.append("  }")
        callexpr = "%s(%s)" % (cname, ", ".join(call))
        if ret != "int":
            body.append("  const char *r_ = %s;" % callexpr)
            body += post[::-1]
            body.append("  (void)cls; return r_ ? (*e)->NewStringUTF(e, r_) : 0;")
        elif jret == "jint":
            body.append("  (void)e; (void)cls; return %s;" % callexpr)
        else:
            body.append("  const int rc_ = %s;" % callexpr)
            body += post[::-1]
            if comm_expr or (cname.startswith("dm_comm_create") or cname == "dm_comm_unique_id"):
                body.append("  (void)cls; if (rc_) raise_comm(e, %s, rc_);" % (comm_expr or "0"))     # communicator calls: dm_comm_last_error
            else:
                body.append("  (void)cls; if (rc_) raise(e, %s, rc_);" % handle_expr)
        body.append("}")
        c.append("\n".join(body))
        sc.append("  @native def %s(%s): %s" % (meth, ", ".join(sparams), sret))
    return c, sc


C_HEAD = '''/* dismember_jni.c — JNI shim over include/dismember_hip.h: one native method of com.mass.hip.Native per C entry point.
 * GENERATED by tools/gen_jni.py from the header — do not edit; `python tools/gen_jni.py --check` (tests/test_jni_shim.py)
 * fails when this file and the header disagree.  Build (needs a JDK; the build image has none, see jni/Makefile):
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include dismember_jni.c -L../dismember_amd -ldismember_hip -o libdismember_jni.so
 * The reference binds its own native math the same way (BigDL MKL JNI, project/Dependencies.scala:27-29). */
#include <jni.h>
#include <stdint.h>

#include "dismember_hip.h"

/* a non-zero status becomes the exception the Scala code threw at that point */
static void raise_msg(JNIEnv *e, int rc, const char *msg) {
  const char *cls = rc == DM_ERR_INDEX ? "java/lang/ArrayIndexOutOfBoundsException"      /* LookupTable.scala:47-53 */
                  : rc == DM_ERR_INVALID ? "java/lang/IllegalArgumentException"           /* require(...) */
                  : rc == DM_ERR_STATE ? "java/lang/IllegalStateException"
                  : rc == DM_ERR_UNSUPPORTED ? "java/lang/UnsupportedOperationException" : "java/lang/RuntimeException";
  (*e)->ThrowNew(e, (*e)->FindClass(e, cls), msg && *msg ? msg : "dismember_hip call failed");
}
static void raise(JNIEnv *e, dm_handle_t h, int rc) { raise_msg(e, rc, dm_last_error(h)); }
/* communicator entry points keep their message in the communicator (NULL: the last create-time error) */
static void raise_comm(JNIEnv *e, dm_comm_t c, int rc) { raise_msg(e, rc, dm_comm_last_error(c)); }
'''

C_DR = '''
/* dm_dr_load_model: the struct carries per-layer arrays; the Scala side passes LayerModel / RerankModel storage arrays
 * (deep-retrieval/.../model/DeepRetrieval.scala:90-106), fp64 like the reference */
JNIEXPORT void JNICALL Java_com_mass_hip_Native_drLoadModelF64(JNIEnv *e, jclass cls, jlong h, jint embed, jint seqLen, jint numNode,
    jint numLayer, jlong numItem, jdoubleArray layerEmb, jobjectArray layerW, jobjectArray layerB, jdoubleArray rerankEmb,
    jdoubleArray rerankW, jdoubleArray rerankB, jdoubleArray softmaxW, jdoubleArray softmaxB) {
  dm_dr_model m = {0};
  const void *w[8] = {0}, *b[8] = {0};
  jdoubleArray wa[8] = {0}, ba[8] = {0};
  if (numLayer < 2 || numLayer > 8) { raise(e, (dm_handle_t)(intptr_t)h, DM_ERR_INVALID); return; }
  m.dtype = DM_F64; m.on_device = 0; m.embed = embed; m.seq_len = seqLen; m.num_node = numNode; m.num_layer = numLayer; m.num_item = numItem;
  for (int d = 0; d < numLayer; d++) {
    wa[d] = (jdoubleArray)(*e)->GetObjectArrayElement(e, layerW, d); ba[d] = (jdoubleArray)(*e)->GetObjectArrayElement(e, layerB, d);
    w[d] = (*e)->GetDoubleArrayElements(e, wa[d], 0); b[d] = (*e)->GetDoubleArrayElements(e, ba[d], 0);
  }
  m.layer_emb = (*e)->GetDoubleArrayElements(e, layerEmb, 0); m.layer_w = w; m.layer_b = b;
  if (rerankEmb) {
    m.rerank_emb = (*e)->GetDoubleArrayElements(e, rerankEmb, 0); m.rerank_w = (*e)->GetDoubleArrayElements(e, rerankW, 0);
    m.rerank_b = (*e)->GetDoubleArrayElements(e, rerankB, 0); m.softmax_w = (*e)->GetDoubleArrayElements(e, softmaxW, 0);
    m.softmax_b = (*e)->GetDoubleArrayElements(e, softmaxB, 0);
  }
  const int rc_ = dm_dr_load_model((dm_handle_t)(intptr_t)h, &m);
  if (rerankEmb) {
    (*e)->ReleaseDoubleArrayElements(e, softmaxB, (jdouble *)m.softmax_b, JNI_ABORT); (*e)->ReleaseDoubleArrayElements(e, softmaxW, (jdouble *)m.softmax_w, JNI_ABORT);
    (*e)->ReleaseDoubleArrayElements(e, rerankB, (jdouble *)m.rerank_b, JNI_ABORT); (*e)->ReleaseDoubleArrayElements(e, rerankW, (jdouble *)m.rerank_w, JNI_ABORT);
    (*e)->ReleaseDoubleArrayElements(e, rerankEmb, (jdouble *)m.rerank_emb, JNI_ABORT);
  }
  (*e)->ReleaseDoubleArrayElements(e, layerEmb, (jdouble *)m.layer_emb, JNI_ABORT);
  for (int d = 0; d < numLayer; d++) {
    (*e)->ReleaseDoubleArrayElements(e, ba[d], (jdouble *)b[d], JNI_ABORT); (*e)->ReleaseDoubleArrayElements(e, wa[d], (jdouble *)w[d], JNI_ABORT);
  }
  (void)cls; if (rc_) raise(e, (dm_handle_t)(intptr_t)h, rc_);
}
'''

S_HEAD = '''// Native.scala — the JNI surface of libdismember_hip.so, one @native method per entry point of include/dismember_hip.h.
// GENERATED by tools/gen_jni.py (do not edit).  Handles, communicators and device pointers are Long; a failing call throws
// the exception the reference's Scala code threw at that point (jni/dismember_jni.c: raise).
package com.mass.hip

object Native {
  System.loadLibrary("dismember_jni")

'''
S_DR = '''  @native def drLoadModelF64(h: Long, embed: Int, seqLen: Int, numNode: Int, numLayer: Int, numItem: Long, layerEmb: Array[Double],
                             layerW: Array[Array[Double]], layerB: Array[Array[Double]], rerankEmb: Array[Double],
                             rerankW: Array[Double], rerankB: Array[Double], softmaxW: Array[Double], softmaxB: Array[Double]): Unit
'''


def render():
    protos = parse_header(open(HDR).read())
    c, sc = gen(protos)
    ctext = C_HEAD + "\n" + "\n\n".join(c) + "\n" + C_DR
    stext = S_HEAD + "\n".join(sc) + "\n" + S_DR + "}\n"
    return protos, ctext, stext


def main():
    protos, ctext, stext = render()
    if "--check" in sys.argv:
        ok = open(OUT_C).read() == ctext and open(OUT_SCALA).read() == stext
        print("jni shim %s" % ("up to date" if ok else "OUT OF DATE: run python tools/gen_jni.py"))
        return 0 if ok else 1
    os.makedirs(os.path.dirname(OUT_C), exist_ok=True)
    os.makedirs(os.path.dirname(OUT_SCALA), exist_ok=True)
    open(OUT_C, "w").write(ctext)
    open(OUT_SCALA, "w").write(stext)
    print("wrote %s (%d entry points) and %s" % (os.path.relpath(OUT_C, ROOT), len(protos), os.path.relpath(OUT_SCALA, ROOT)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
