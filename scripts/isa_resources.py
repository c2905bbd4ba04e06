#!/usr/bin/env python
"""Per-kernel ISA resources of the shipped gfx950 code objects -> CSV (profiles/rNN/isa_resources.csv).

For every object file under dsopp_amd/lib/ the gfx950 code object is taken out of the HIP fat binary (llvm-objdump --offloading,
on a copy in a temporary directory), its kernel descriptors are read from the AMDGPU metadata note (llvm-readelf --notes: VGPR /
AGPR / SGPR counts, static LDS, scratch bytes per lane, VGPR / SGPR spill counts, max workgroup size) and the disassembly
(llvm-objdump -d) is scanned per kernel symbol for FLAT, scratch and MFMA instructions and the total instruction count.
Runs without a GPU.

    python scripts/isa_resources.py [--out profiles/r03/isa_resources.csv] [--filter sweep]
"""
import argparse
import csv
import glob
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                         input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}


def short(name):
    """kernel name without namespaces / argument list; template arguments kept"""
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].replace("dsopp_hip::", "")


def code_object(obj, tmp):
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True)
    cands = glob.glob(local + ".*gfx950*")
    return cands[0] if cands else None


def kernel_notes(co):
    text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .agpr_count") or (line.lstrip().startswith("- .") and key in ("agpr_count", "args")):
            if key == "agpr_count":
                cur = {}
                kernels.append(cur)
        if cur is None:
            continue
        if key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count",
                   "sgpr_spill_count", "max_flat_workgroup_size", "kernarg_segment_size"):
            cur[key] = int(val)
        elif key == "name" and "name" not in cur and val.startswith("_Z"):
            cur["name"] = val
        elif key == "symbol":
            cur["symbol"] = val
    return [k for k in kernels if "name" in k]


def disassembly_counts(co):
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    counts, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = counts.setdefault(m.group(1), dict(instructions=0, flat=0, scratch=0, mfma=0, readlane=0, s_swappc=0))
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if not m:
            continue
        op = m.group(1)
        cur["instructions"] += 1
        if op.startswith("flat_"):
            cur["flat"] += 1
        elif op.startswith("scratch_"):
            cur["scratch"] += 1
        elif op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_writelane"):
            cur["readlane"] += 1
        elif op.startswith("s_swappc"):
            cur["s_swappc"] += 1
    return counts


def waves_per_simd(vgpr, agpr):
    """gfx950: 512 unified VGPR+AGPR per lane and SIMD, allocation granule 8, at most 8 waves"""
    total = max(1, ((vgpr + agpr + 7) // 8) * 8)
    return min(8, 512 // total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--filter", default=None, help="substring of the demangled kernel name")
    ap.add_argument("--lib", default="lib", help="directory under dsopp_amd/ holding the object files")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "dsopp_amd", args.lib, "*.o"))):
            co = code_object(obj, tmp)
            if not co:
                continue
            notes = kernel_notes(co)
            counts = disassembly_counts(co)
            names = demangle([k["name"] for k in notes])
            # non-kernel device functions (out-of-line callees) of this object: their FLAT / scratch instructions belong to no kernel row
            kernel_syms = {k["name"] for k in notes}
            for k in notes:
                c = counts.get(k["name"], {})
                rows.append(dict(object=os.path.basename(obj), kernel=short(names[k["name"]]), vgpr=k.get("vgpr_count", 0), agpr=k.get("agpr_count", 0),
                                 sgpr=k.get("sgpr_count", 0), waves_per_simd=waves_per_simd(k.get("vgpr_count", 0), k.get("agpr_count", 0)),
                                 lds_static_bytes=k.get("group_segment_fixed_size", 0), scratch_bytes_per_lane=k.get("private_segment_fixed_size", 0),
                                 vgpr_spills=k.get("vgpr_spill_count", 0), sgpr_spills=k.get("sgpr_spill_count", 0),
                                 max_workgroup=k.get("max_flat_workgroup_size", 0), instructions=c.get("instructions", 0), flat=c.get("flat", 0),
                                 scratch_instr=c.get("scratch", 0), mfma=c.get("mfma", 0), lane_moves=c.get("readlane", 0), calls=c.get("s_swappc", 0)))
            for sym, c in counts.items():
                if sym in kernel_syms or sym.endswith(".kd") or not sym.startswith("_Z"):
                    continue
                nm = short(demangle([sym])[sym])
                rows.append(dict(object=os.path.basename(obj), kernel="(device function) " + nm, vgpr="", agpr="", sgpr="", waves_per_simd="",
                                 lds_static_bytes="", scratch_bytes_per_lane="", vgpr_spills="", sgpr_spills="", max_workgroup="",
                                 instructions=c["instructions"], flat=c["flat"], scratch_instr=c["scratch"], mfma=c["mfma"], lane_moves=c["readlane"],
                                 calls=c["s_swappc"]))
    if args.filter:
        rows = [r for r in rows if args.filter in r["kernel"]]
    fields = list(rows[0].keys()) if rows else []
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=fields)
            w.writeheader()
            w.writerows(rows)
        print(f"{len(rows)} rows -> {args.out}")
        tot = {k: sum(int(r[k] or 0) for r in rows) for k in ("flat", "scratch_instr", "vgpr_spills", "sgpr_spills")}
        print("totals:", tot)
    else:
        print(",".join(fields))
        for r in rows:
            print(",".join(str(r[f]) for f in fields))


if __name__ == "__main__":
    main()
