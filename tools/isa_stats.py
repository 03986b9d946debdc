#!/usr/bin/env python3
"""Static ISA statistics of the library's kernels, without a GPU: takes the gfx950 code objects out of the built
library (one per translation unit) and reports, per kernel whose demangled name matches the regular expression, the register
budget (from the code object's metadata notes) and the instruction mix (from the disassembly).

    python tools/isa_stats.py 'step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>' [--keep /tmp/isa]

Used to check that a change to shared device code leaves the benchmarked instantiations alone (VGPRs, VALU count) and to
count what a kernel issues per wave."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(out_dir):
    """Every gfx950 code object of the built library (one per translation unit: csrc/kernels_*.hip, mbt_env.hip), taken out of
    libmbtenv.so the way tools/dbg/kernel_resources.py does - the library is what runs, and build.py keeps it current."""
    from mbt_gym_amd import build as b

    lib = b.build_native()
    fat = os.path.join(out_dir, "fatbin.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(out_dir, "unused.so")], check=True, capture_output=True)
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for k, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, elf = os.path.join(out_dir, f"bundle{k}.bin"), os.path.join(out_dir, f"unit{k}.gfx950.o")
        open(piece, "wb").write(blob[lo:hi])
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={elf}"], check=True)
        out.append(elf)
    return out


def kernel_metadata(co):
    """{mangled name: {vgpr, sgpr, agpr, lds, scratch, spill}} from the msgpack notes, as llvm-readelf prints them."""
    text = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, {}
    for line in text.splitlines():
        m = re.match(r"^  (- | {2})\.(\w+):\s*(.*)", line)  # kernel-level keys only (argument records are indented further)
        if not m:
            continue
        if m.group(1) == "- ":
            cur = {}
        key, val = m.group(2), m.group(3).strip()
        cur[key] = val
        if key == "name" and val:
            out[val.strip("'\"")] = cur
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def instruction_mix(co, symbol):
    text = subprocess.run([LLVM + "/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", co], check=True, capture_output=True, text=True).stdout
    mix = {"valu": 0, "salu": 0, "smem": 0, "vmem_load": 0, "vmem_store": 0, "lds": 0, "trans": 0, "f64": 0, "pk": 0, "mfma": 0, "branch": 0, "total": 0}
    for line in text.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not m:
            continue
        op = m.group(1)
        mix["total"] += 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            mix["mfma"] += 1
        elif op.startswith("v_"):
            mix["valu"] += 1
            if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
                mix["trans"] += 1
            if "_f64" in op:
                mix["f64"] += 1
            if op.startswith("v_pk_"):
                mix["pk"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            mix["smem"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            mix["branch"] += 1
        elif op.startswith("s_"):
            mix["salu"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
            mix["vmem_load"] += 1
        elif op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("scratch_store") or op.startswith("global_atomic"):
            mix["vmem_store"] += 1
        elif op.startswith("ds_"):
            mix["lds"] += 1
    return mix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", help="regular expression on the demangled kernel name")
    ap.add_argument("--keep", default="/tmp/isa", help="directory for the device-only code object")
    args = ap.parse_args()
    os.makedirs(args.keep, exist_ok=True)
    rx = re.compile(args.pattern)
    for co in code_objects(args.keep):
        meta = kernel_metadata(co)
        names = demangle(list(meta))
        for mangled, pretty in sorted(names.items(), key=lambda kv: kv[1]):
            if not rx.search(pretty):
                continue
            m, mix = meta[mangled], instruction_mix(co, mangled)
            print(pretty)
            print("   vgpr %s agpr %s sgpr %s lds %s scratch %s vgpr_spill %s" % (m.get("vgpr_count"), m.get("agpr_count"), m.get("sgpr_count"),
                                                                                  m.get("group_segment_fixed_size"), m.get("private_segment_fixed_size"), m.get("vgpr_spill_count")))
            print("   " + " ".join(f"{k} {v}" for k, v in mix.items()))


if __name__ == "__main__":
    main()
