"""The JNI shim and the Scala side exist as FILES and cannot drift from the C header:
  * jni/dismember_jni.c and scala/com/mass/hip/Native.scala are exactly what tools/gen_jni.py derives from
    include/dismember_hip.h;
  * every entry point of the ABI (dismember_amd/_native.SIGNATURES) has its native method(s) on both sides;
  * the shim type-checks as C (against a minimal test-only stand-in for <jni.h>: the image has no JDK; with JAVA_HOME set the
    real header is used instead);
  * every Native.<method>(...) the hand-written Scala facades call exists with that number of arguments.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_generated_files_are_current():
    import gen_jni
    _, ctext, stext = gen_jni.render()
    assert open(gen_jni.OUT_C).read() == ctext, "jni/dismember_jni.c is stale: run python tools/gen_jni.py"
    assert open(gen_jni.OUT_SCALA).read() == stext, "scala/com/mass/hip/Native.scala is stale: run python tools/gen_jni.py"


def _native_methods():
    text = open(os.path.join(ROOT, "scala", "com", "mass", "hip", "Native.scala")).read()
    out = {}
    for m in re.finditer(r"@native def (\w+)\((.*?)\): (\w+)", text, flags=re.S):
        args = [a for a in m.group(2).replace("\n", " ").split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def test_every_abi_entry_point_is_bound():
    import gen_jni
    from dismember_amd import _native as N
    methods = _native_methods()
    csrc = open(os.path.join(ROOT, "jni", "dismember_jni.c")).read()
    for name in N.SIGNATURES:
        base = gen_jni.camel(name)
        hits = [m for m in methods if m == base or (m.startswith(base) and m[len(base):] in ("F32", "F64"))]
        assert hits, name
        for m in hits:
            assert "Java_com_mass_hip_Native_%s(" % m.replace("_", "_1") in csrc, m
    assert len(methods) >= len(N.SIGNATURES)


def test_shim_type_checks_as_c():
    java_home = os.environ.get("JAVA_HOME", "")
    if java_home and os.path.exists(os.path.join(java_home, "include", "jni.h")):
        inc = ["-I" + os.path.join(java_home, "include"), "-I" + os.path.join(java_home, "include", "linux")]
    else:
        inc = ["-I" + os.path.join(ROOT, "tests", "cpp", "jni_stub")]
    subprocess.check_call(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include")] + inc +
                          [os.path.join(ROOT, "jni", "dismember_jni.c")])


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_scala_facades_call_existing_natives():
    methods = _native_methods()
    d = os.path.join(ROOT, "scala", "com", "mass", "hip")
    facades = [f for f in os.listdir(d) if f.endswith(".scala") and f != "Native.scala"]
    assert sorted(facades) == ["DeepRetrieval.scala", "HipEngine.scala", "JTM.scala", "LocalOptimizer.scala", "OTM.scala", "OTMLocalOptimizer.scala", "TDM.scala"]
    calls = 0
    for f in facades:
        text = open(os.path.join(d, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"Native\.(\w+)\(", text):
            name = m.group(1)
            assert name in methods, (f, name)
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            assert len(_split_args(text[m.end():i - 1])) == methods[name], (f, name)
            calls += 1
    assert calls >= 20


def test_integration_excerpts_match_sources():
    """INTEGRATION.md shows the binding a maintainer adds; its code blocks are excerpts of the generated / hand-written sources
    and must stay literal (a changed argument order in the generator must not leave a stale snippet in the document)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"<!-- excerpt: (\S+) -->\n```\w*\n(.*?)\n```", text, flags=re.S)
    assert len(blocks) >= 4
    for path, body in blocks:
        src = open(os.path.join(ROOT, path)).read()
        assert body in src, "INTEGRATION.md excerpt of %s is stale" % path
