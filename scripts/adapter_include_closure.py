#!/usr/bin/env python
"""Why the reference-side adapters (dsopp_amd/host/reference_adapter/*.hpp) cannot be put through a compiler in this image — as a list, not
a claim: the transitive include closure of the two base-class headers they derive from, inside the reference tree, and for every
third-party header the closure needs (none of which exists in the image) the reference headers that include it directly and the
third-party types those headers use in DECLARATIONS (class members, function signatures, aliases) — i.e. what a "signature-only" stand-in
would have to define for the base classes to parse at all.

    python scripts/adapter_include_closure.py [/root/reference]      (prints a markdown table; INTEGRATION.md §5 holds a copy)"""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
BASES = ["src/energy/problems/include/energy/problems/photometric_bundle_adjustment/photometric_bundle_adjustment.hpp",
         "src/energy/problems/include/energy/problems/pose_alignment/pose_alignment.hpp"]
NAMESPACES = {"Eigen": r"Eigen::(\w+)", "sophus": r"Sophus::(\w+)", "ceres": r"ceres::(\w+)", "opencv2": r"cv::(\w+)", ".pb.h": r"proto::(\w+)"}


def main():
    roots = [d for d, _, _ in os.walk(os.path.join(REF, "src")) if d.endswith("/include") or d.endswith("/internal")]

    def find(inc):
        for r in roots:
            p = os.path.join(r, inc)
            if os.path.exists(p):
                return p
        return None

    seen, third = {}, {}

    def walk(p):
        if p in seen:
            return
        seen[p] = open(p, errors="ignore").read()
        for m in re.finditer(r'^\s*#include\s*[<"]([^>"]+)[>"]', seen[p], re.M):
            inc = m.group(1)
            q = find(inc)
            if q:
                walk(q)
            elif "/" in inc or "." in inc:
                third.setdefault(inc, []).append(os.path.relpath(p, REF))

    for b in BASES:
        walk(os.path.join(REF, b))
    print(f"{len(seen)} reference headers in the closure of the two base classes; {len(third)} third-party headers, none in the image:\n")
    print("| third-party header | included directly by | its types in declarations of the closure |")
    print("|---|---|---|")
    for inc in sorted(third):
        pat = next((v for k, v in NAMESPACES.items() if k in inc), None)
        used = set()
        if pat:
            for text in seen.values():
                used.update(re.findall(pat, text))
        users = ", ".join("`" + os.path.basename(u) + "`" for u in sorted(set(third[inc])))
        print(f"| `{inc}` | {users} | {', '.join(sorted(used)[:14]) + (' …' if len(used) > 14 else '')} |")


if __name__ == "__main__":
    main()
