#!/usr/bin/env python3
"""Timing-experiment variants of the shipped kernels, kept OUT of the shipped sources.

The kernels under sinnerf_amd/csrc/ used to carry `#ifdef SN_ABL_* / SN_T_* / SN_V3_*` branches that leave work out (wrong results
by design) so that a launch could be timed without its stores / staging / trunk ...  They do not belong in the product: this
tool owns them now.  tools/ablations.json lists, per experiment macro, the text edits that re-create the experiment from the
CLEAN source (`old` must occur exactly once in the file -- a kernel edit that moves it fails loudly here, not silently in a
profile); `build` applies the edits of the requested macros to a scratch copy of csrc/ under build/variants/<name>/ and builds
lib_<name>.so there (same ABI; load it with SINNERF_HIP_LIB=...).

  tools/ablate.py build NAME MACRO[=VALUE] ... [-- make-variable=value ...]
  tools/ablate.py list
  tools/ablate.py strip            (one-off, already run: unifdef the macros out of csrc/ and write tools/ablations.json)
"""
import json
import os
import re
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(R, "sinnerf_amd", "csrc")
RULES = os.path.join(R, "tools", "ablations.json")
MACROS = ["SN_DW_ONLY_VARIANT", "SN_ABL_NO_STAGE", "SN_ABL_NO_G_STORE", "SN_ABL_NO_MASK_LOAD", "SN_ABL_NO_EMB_STORE",
          "SN_ABL_NO_ACTS_STORE", "SN_ABL_NO_STATE_STORE", "SN_T_DEPHASE", "SN_T_ABL_NO_RAYLOAD", "SN_T_NO_EMB", "SN_T_SKIP_TRUNK",
          "SN_T_SWAP_REV", "SN_T_ABL_NO_DIRSTORE", "SN_V3_SKIP_TRUNK", "SN_V3_DMA_WAVE0", "SN_DWN_NO_QUAD"]


COND = re.compile(r"\s*#\s*(ifdef|ifndef|if)\s+(?:defined\()?(\w+)\)?\s*(//.*)?$")


def process(src):
    """resolve every conditional on one of MACROS as 'macro undefined' (0 for `#if MACRO`).  Returns the kept lines and the edits
    (macro, position in the kept lines, number of kept lines the edit replaces, replacement lines) that bring the 'defined' branch
    back.  Conditionals on other macros are left alone; nested experiment macros are resolved first."""
    out, edits, i = [], [], 0
    while i < len(src):
        m = COND.match(src[i])
        if not (m and m.group(2) in MACROS):
            out.append(src[i])
            i += 1
            continue
        kind, macro = m.group(1), m.group(2)
        depth, j, then, els = 1, i + 1, [], []
        cur = then
        while True:
            ln = src[j]
            if re.match(r"\s*#\s*(ifdef|ifndef|if)\b", ln):
                depth += 1
            elif re.match(r"\s*#\s*endif\b", ln):
                depth -= 1
                if depth == 0:
                    break
            elif re.match(r"\s*#\s*else\b", ln) and depth == 1:
                cur = els
                j += 1
                continue
            cur.append(ln)
            j += 1
        keep, drop = (then, els) if kind == "ifndef" else (els, then)
        keep, sub = process(keep)
        drop, _ = process(drop)
        edits.append((macro, len(out), len(keep), drop))
        edits.extend((mc, len(out) + pos, n, new) for mc, pos, n, new in sub)
        out.extend(keep)
        i = j + 1
    return out, edits


def strip():
    rules = {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".h")):
            continue
        p = os.path.join(CSRC, f)
        out, edits = process(open(p).read().split("\n"))
        if not edits:
            continue
        text = "\n".join(out)
        for macro, pos, n, new in edits:
            for ctx in range(1, 12):                      # grow the leading context until the anchor is unique in the clean file
                old = "\n".join(out[pos - ctx:pos + n])
                if text.count(old) == 1:
                    break
            else:
                raise SystemExit("no unique anchor for %s in %s" % (macro, f))
            rules.setdefault(macro, []).append({"file": f, "old": old, "new": "\n".join(out[pos - ctx:pos] + new)})
        open(p, "w").write(text)
    if not rules:
        raise SystemExit("csrc/ carries none of the experiment macros (already stripped): tools/ablations.json left alone")
    json.dump(rules, open(RULES, "w"), indent=1)
    print({k: len(v) for k, v in rules.items()})


def build(name, macros, make_vars):
    rules = json.load(open(RULES))
    dst = os.path.join(R, "build", "variants", name)
    if os.path.exists(dst):
        shutil.rmtree(dst)
    os.makedirs(os.path.join(dst, "sinnerf_amd"))
    shutil.copytree(CSRC, os.path.join(dst, "sinnerf_amd", "csrc"),
                    ignore=shutil.ignore_patterns("*.o", "*.so", "*.checked", "*.s"))
    for sub in ("tools", "include"):
        os.symlink(os.path.join(R, sub), os.path.join(dst, sub))
    cflags = []
    for mv in macros:
        macro, _, val = mv.partition("=")
        assert macro in rules, "unknown experiment %s (tools/ablate.py list)" % macro
        for r in rules[macro]:
            p = os.path.join(dst, "sinnerf_amd", "csrc", r["file"])
            text = open(p).read()
            assert text.count(r["old"]) == 1, "%s: the anchor of %s no longer occurs exactly once in %s" % (name, macro, r["file"])
            open(p, "w").write(text.replace(r["old"], r["new"]))
        cflags.append("-D%s%s" % (macro, "=" + val if val else ""))
    env = dict(os.environ)
    cmd = ["make", "-C", os.path.join(dst, "sinnerf_amd", "csrc"), "-j8", "libsinnerf_hip.so", "EXTRA_CXXFLAGS=" + " ".join(cflags)] + make_vars
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL)
    out = os.path.join(R, "build", "variants", "lib_%s.so" % name)
    shutil.copy(os.path.join(dst, "sinnerf_amd", "csrc", "libsinnerf_hip.so"), out)
    print("built", out)


if __name__ == "__main__":
    if sys.argv[1] == "strip":
        strip()
    elif sys.argv[1] == "list":
        for k, v in json.load(open(RULES)).items():
            print("%-24s %s" % (k, sorted({r["file"] for r in v})))
    else:
        args = sys.argv[3:]
        mk = args[args.index("--") + 1:] if "--" in args else []
        args = args[:args.index("--")] if "--" in args else args
        build(sys.argv[2], args, mk)
