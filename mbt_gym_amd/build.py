"""Builds libmbtenv.so (HIP, gfx950 only) in-tree.  hipcc cross-compiles without a GPU present.

Staleness is decided by CONTENT, not by time stamps: the sha256 of every source that goes into the library
(csrc/*.hip, csrc/*.hpp, include/mbt_env.h, in name order) is baked into it (`mbt_source_hash()`), and a library
whose hash differs from the sources next to it is rebuilt (here) or refused (`_native.load_library`) - a stale
prebuilt .so is not a silent failure mode.
"""
import glob
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
HEADER = os.path.join(PKG_DIR, "..", "include", "mbt_env.h")
LIB_PATH = os.path.join(PKG_DIR, "libmbtenv.so")
SOURCES = ["mbt_env.hip"]
# -ffp-contract=off: the step kernel and the fused rollout kernel inline the same arithmetic and must agree bit for bit;
# letting the compiler pick FMA contractions per kernel breaks that (1-ulp reward differences were observed).
# RCCL is bound with dlopen at run time (csrc/mbt_env.hip: rccl()), hence -ldl and no -lrccl.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
LINK_FLAGS = ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]


def source_files():
    """Everything the library is compiled from, in a fixed order."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")))
    return files + [os.path.normpath(HEADER)]


def source_hash() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode())
        h.update(b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


HASH_MARKER = b"mbt-source-hash:"


def library_hash(path: str = LIB_PATH):
    """The hash baked into an existing library (read from the file, without loading it), or None."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        blob = f.read()
    at = blob.find(HASH_MARKER)
    if at < 0:
        return None
    start = at + len(HASH_MARKER)
    return blob[start:start + 64].decode("ascii", "replace")


def _stale() -> bool:
    return library_hash() != source_hash()


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension if it is missing or was built from different sources; returns the .so path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmbtenv.so cannot be built (and there is no CPU fallback)")
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + [f'-DMBT_SOURCE_HASH="{source_hash()}"'] + [os.path.join(CSRC, s) for s in SOURCES] + LINK_FLAGS + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)  # a process that has the old library mapped keeps its (unlinked) file
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
