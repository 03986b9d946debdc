#!/usr/bin/env python3
"""Static resources of the kernels in mbt_gym_amd/libmbtenv.so, read from the code object's own metadata (no GPU): registers,
spills, LDS, code size, and the occupancy they allow on gfx950 (512 VGPRs per SIMD lane, allocation granule 8; 8 waves at most).

    python tools/dbg/kernel_resources.py [substring ...]      # default: the kernels bench.py times + the speed family

Unbundles the gfx950 code object with clang-offload-bundler and reads NT_AMDGPU_METADATA with llvm-readelf; code sizes from the
symbol table.  What `rocprofv3 --kernel-trace` shows as a kernel's name is the demangled `.name` below.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "mbt_gym_amd", "libmbtenv.so")


def code_objects(workdir):
    """Every gfx950 code object in the library: one offload bundle per translation unit (csrc/kernels_*.hip, mbt_env.hip), concatenated in
    .hip_fatbin - split at the bundler's magic."""
    fat = os.path.join(workdir, "fatbin.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(workdir, "unused.so")], check=True, capture_output=True)
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    objects = []
    for k, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, obj = os.path.join(workdir, f"bundle{k}.bin"), os.path.join(workdir, f"gfx950_{k}.co")
        open(piece, "wb").write(blob[lo:hi])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={obj}"],
                       check=True, capture_output=True)
        objects.append(obj)
    return objects


def kernels(obj):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", obj], check=True, capture_output=True, text=True).stdout
    sizes = {}
    for line in subprocess.run([f"{LLVM}/llvm-readelf", "--symbols", "--wide", obj], check=True, capture_output=True, text=True).stdout.splitlines():
        parts = line.split()
        if len(parts) >= 8 and parts[3] == "FUNC":
            sizes[parts[7]] = int(parts[2])
    out = []
    for block in notes.split("  - .agpr_count:")[1:]:
        field = lambda key: re.search(rf"\.{key}:\s+(\S+)", block).group(1)  # noqa: E731
        name = field("name")
        out.append(dict(name=name, vgpr=int(field("vgpr_count")), agpr=int(block.split()[0]), sgpr=int(field("sgpr_count")), lds=int(field("group_segment_fixed_size")),
                        scratch=int(field("private_segment_fixed_size")), vgpr_spill=int(field("vgpr_spill_count")), sgpr_spill=int(field("sgpr_spill_count")),
                        code=sizes.get(name, 0)))
    return out


def demangle(names):
    text = subprocess.run(["c++filt"], input="\n".join(names), check=True, capture_output=True, text=True).stdout
    return text.splitlines()


def waves_per_simd(vgpr, agpr):
    regs = -(-(vgpr + agpr) // 8) * 8  # unified register file, granule 8
    return max(1, min(8, 512 // max(regs, 8)))


DEFAULT = ("step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>", "step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::quadratic>",
           "step_kernel<mbt::Variant<mbt::shape::hawkes", "step_kernel<mbt::Variant<mbt::shape::limit_and_market, mbt::shape::brownian, mbt::shape::pnl>",
           "step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::pnl, mbt::shape::precise>", "speed_step_kernel<mbt::SpeedVariant<")


def main():
    wanted = tuple(sys.argv[1:]) or DEFAULT
    with tempfile.TemporaryDirectory() as workdir:
        rows = [row for obj in code_objects(workdir) for row in kernels(obj)]
    for row, name in zip(rows, demangle([r["name"] for r in rows])):
        row["demangled"] = re.sub(r"^void ", "", name).split("(")[0]
    rows = [r for r in rows if any(w in r["demangled"] for w in wanted)]
    print(f"{len(rows)} kernels of {os.path.relpath(LIB, ROOT)} (gfx950 code object metadata)")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'waves/SIMD':>10} {'LDS B':>7} {'scratch B':>9} {'spill v/s':>9} {'code B':>7}  kernel")
    for r in sorted(rows, key=lambda r: r["demangled"]):
        print(f"{r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {waves_per_simd(r['vgpr'], r['agpr']):10d} {r['lds']:7d} {r['scratch']:9d} "
              f"{str(r['vgpr_spill']) + '/' + str(r['sgpr_spill']):>9} {r['code']:7d}  {r['demangled']}")


if __name__ == "__main__":
    main()
