#!/usr/bin/env python
"""Lint of the reference-side adapters (dsopp_amd/host/reference_adapter/*.{hpp,cpp}) — NOT a compiler.

The adapters can only be compiled inside a DSOPP build tree (Eigen, Sophus, glog, OpenCV, TBB, Ceres; none of them is in this
image, and stand-in headers for template libraries of that size would check the stand-ins, not the adapters).  What can be
checked mechanically, and is checked here, are the three classes of slip a compiler would catch first:

  1. C-ABI call sites: every dsopp_hip_* identifier the adapters use is declared in include/dsopp_hip.h and is called with the
     declared number of arguments;
  2. includes: every #include "..." of the adapters names a header that exists under /root/reference/src (when the reference tree
     is present) or next to the adapter;
  3. reference members: every member the adapters touch on the reference's own objects (keyframe, local frame, landmark, residual
     point, camera mask, pixel map, trust-region options, base class) is spelled as some declaration in the reference header that
     defines that type.

Exit code 0 = clean.  tests/test_host_adapter.py runs it (part 2 and 3 are skipped without /root/reference)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "dsopp_amd", "host", "reference_adapter")
REF = os.environ.get("DSOPP_REFERENCE_ROOT", "/root/reference")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_args(s):
    """top-level comma split of an argument list (no outer parentheses)"""
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<" and not (ch == "<" and depth == 0 and not re.search(r"[A-Za-z_>]\s*$", cur)):
            depth += 1
        elif ch in ")]}>" and depth > 0:
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return args


def matching_paren(text, open_idx):
    depth = 0
    for i in range(open_idx, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i
    return -1


def header_prototypes():
    text = strip_comments(open(os.path.join(ROOT, "include", "dsopp_hip.h")).read())
    protos = {}
    for m in re.finditer(r"\b(dsopp_hip_[a-z0-9_]+)\s*\(", text):
        name, start = m.group(1), m.end() - 1
        end = matching_paren(text, start)
        inner = text[start + 1:end].strip()
        protos[name] = 0 if inner in ("", "void") else len(split_args(inner))
    enums = set(re.findall(r"\b(DSOPP_HIP_[A-Z0-9_]+)\b", text))
    types = set(re.findall(r"\b(dsopp_hip_[a-z0-9_]+)\b", text))
    return protos, enums, types


def check_cabi(files, errors):
    protos, enums, types = header_prototypes()
    for path in files:
        text = strip_comments(open(path).read())
        for m in re.finditer(r"\b(dsopp_hip_[a-z0-9_]+)\s*\(", text):
            name, start = m.group(1), m.end() - 1
            if name not in protos:
                errors.append(f"{os.path.basename(path)}: call of undeclared {name}")
                continue
            end = matching_paren(text, start)
            inner = text[start + 1:end].strip()
            n = 0 if inner == "" else len(split_args(inner))
            if n != protos[name]:
                errors.append(f"{os.path.basename(path)}: {name} called with {n} arguments, declared with {protos[name]}")
        for name in set(re.findall(r"\b(dsopp_hip_[a-z0-9_]+)\b", text)):
            if name not in types:
                errors.append(f"{os.path.basename(path)}: unknown C-ABI identifier {name}")
        for name in set(re.findall(r"\b(DSOPP_HIP_[A-Z0-9_]+)\b", text)):
            if name not in enums and not name.endswith("_HPP") and name != "DSOPP_HIP_CHECKED":
                errors.append(f"{os.path.basename(path)}: unknown C-ABI constant {name}")


def reference_headers():
    out = {}
    for base, _, names in os.walk(os.path.join(REF, "src")):
        for n in names:
            if n.endswith((".hpp", ".h")):
                out.setdefault(n, []).append(os.path.join(base, n))
    return out


def check_includes(files, errors):
    for path in files:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
            if os.path.exists(os.path.join(ADAPTER, os.path.basename(inc))):
                continue
            # the reference's include roots are .../<module>/include/ and .../<module>/internal/
            hits = [p for p in reference_headers().get(os.path.basename(inc), []) if p.replace(os.sep, "/").endswith("/" + inc)]
            if not hits:
                errors.append(f"{os.path.basename(path)}: #include \"{inc}\" matches no header under {REF}/src")


# object expression (regex on the adapter text) -> reference header (path suffix) whose declarations its members must match
MEMBER_RULES = [
    # (a variable called `frame` is a track::ActiveKeyframe in the bundle-adjustment adapter and a LocalFrame in a helper of the aligner)
    (r"\bframe\.", ["track/frames/active_keyframe.hpp", "track/frames/keyframe.hpp", "track/frames/frame.hpp", "track/frames/slam_internal_tracking_frame.hpp",
                    "track/frames/tracking_frame.hpp", "photometric_bundle_adjustment/local_frame.hpp"]),
    (r"\blocal_frame(?:->|\.)", ["photometric_bundle_adjustment/local_frame.hpp"]),
    (r"\b(?:reference_frame|target_frame|reference|target)(?:->|\.)", ["photometric_bundle_adjustment/local_frame.hpp"]),
    (r"\blandmark\.", ["photometric_bundle_adjustment/local_frame.hpp"]),
    (r"\bpoint_residuals\[[^\]]+\]\.", ["photometric_bundle_adjustment/local_frame.hpp"]),
    (r"\btrust_region_options\.", ["photometric_bundle_adjustment/trust_region_photometric_bundle_adjustment_options.hpp"]),
    (r"\bthis->", ["photometric_bundle_adjustment/photometric_bundle_adjustment.hpp", "pose_alignment/pose_alignment.hpp"]),
    (r"\bmask\.", ["mask/camera_mask.hpp"]),
    (r"\blevel\.", ["features/camera/pixel_map.hpp"]),
    (r"\bmodel\.", ["camera_model/pinhole/pinhole_camera.hpp", "camera_model/camera_model_base.hpp"]),
]


def check_members(files, errors):
    headers = reference_headers()

    def load(suffixes):
        text = ""
        for suf in suffixes:
            for p in headers.get(os.path.basename(suf), []):
                if p.replace(os.sep, "/").endswith(suf):
                    text += open(p).read()
        return text

    cache = {}
    for path in files:
        text = strip_comments(open(path).read())
        for pattern, suffixes in MEMBER_RULES:
            key = tuple(suffixes)
            if key not in cache:
                cache[key] = load(suffixes)
            ref_text = cache[key]
            if not ref_text:
                errors.append(f"reference header(s) {suffixes} not found under {REF}/src")
                continue
            for m in re.finditer(pattern + r"(?:template\s+)?([A-Za-z_][A-Za-z0-9_]*)", text):
                member = m.group(1)
                if not re.search(r"\b" + re.escape(member) + r"\b", ref_text):
                    errors.append(f"{os.path.basename(path)}: `{m.group(0)}`: no declaration of `{member}` in {suffixes[0]} (+{len(suffixes) - 1} more)")


def main():
    files = sorted(os.path.join(ADAPTER, n) for n in os.listdir(ADAPTER) if n.endswith((".hpp", ".cpp")))
    errors = []
    check_cabi(files, errors)
    have_ref = os.path.isdir(os.path.join(REF, "src"))
    if have_ref:
        check_includes(files, errors)
        check_members(files, errors)
    for e in sorted(set(errors)):
        print("adapter_lint:", e)
    print(f"adapter_lint: {len(files)} files, C-ABI calls{' , includes, reference members' if have_ref else ' only (no reference tree)'}: "
          f"{'clean' if not errors else str(len(set(errors))) + ' finding(s)'}")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
