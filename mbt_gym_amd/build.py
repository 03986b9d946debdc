"""Builds libmbtenv.so (HIP, gfx950 only) in-tree.  hipcc cross-compiles without a GPU present.

Staleness is decided by CONTENT, not by time stamps: the sha256 of every source that goes into the library
(csrc/*.hip, csrc/*.hpp, include/mbt_env.h, in name order) is baked into it (`mbt_source_hash()`), and a library
whose hash differs from the sources next to it is rebuilt (here) or refused (`_native.load_library`) - a stale
prebuilt .so is not a silent failure mode.
"""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
HEADER = os.path.join(PKG_DIR, "..", "include", "mbt_env.h")
LIB_PATH = os.path.join(PKG_DIR, "libmbtenv.so")
# mbt_env.hip: the C ABI, the helper kernels and the run-time compiler; kernels_*.hip: the step / rollout kernel instantiations,
# one family per translation unit (csrc/kernel_table.hpp) - compiled in parallel, each object cached under build/ by the
# content of what it was compiled from.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function",
               "-mllvm", "-amdgpu-mfma-vgpr-form"]
# (experiments: MBT_EXTRA_HIPCC_FLAGS="-DMBT_SPEED_PRECISE_GROUPS=2 ..." compiles a variant; the object cache is keyed by the flags)
HIPCC_FLAGS += os.environ.get("MBT_EXTRA_HIPCC_FLAGS", "").split()
LINK_FLAGS = ["-shared", "-fPIC", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
# object cache: beside the sources in a checkout; under the user's cache directory when the package is installed (site-packages is not
# the place for build products, and may not be writable)
_IN_CHECKOUT = os.path.isdir(os.path.join(PKG_DIR, "..", ".git")) or os.access(os.path.join(PKG_DIR, ".."), os.W_OK)
OBJ_DIR = (os.path.join(PKG_DIR, "..", "build", "libmbtenv") if _IN_CHECKOUT
           else os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.expanduser("~/.cache")), "mbt_gym_amd", "libmbtenv"))

# Sanitizer builds of the HOST side (README "Sanitizers"): MBT_SANITIZE=address,undefined or MBT_SANITIZE=thread
# compiles mbt_env.hip - the C ABI: every spin flag, mapped buffer and per-thread state of the library lives there - with
# -fsanitize=... for the host only (-fno-gpu-sanitize: device code is untouched, GPU sanitizers are not available on the pool) and
# links a SEPARATE library, libmbtenv.<asan|tsan>.so, next to the production one; _native.load_library loads it when
# MBT_LIBRARY_VARIANT names it (the sanitizer's runtime has to be LD_PRELOADed into the uninstrumented python: sanitizer_runtime()).
SANITIZE = os.environ.get("MBT_SANITIZE", "").strip()
VARIANT = {"": "", "address,undefined": "asan", "address": "asan", "undefined": "asan", "thread": "tsan"}.get(SANITIZE)
if VARIANT is None:
    raise RuntimeError(f"MBT_SANITIZE={SANITIZE!r}: use address,undefined or thread")
SANITIZE_FLAGS = [f"-fsanitize={SANITIZE}", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"] if SANITIZE else []
# Experiments (tools/dbg/build_variant.py): MBT_BUILD_VARIANT=name with MBT_EXTRA_HIPCC_FLAGS="-D..." builds libmbtenv.<name>.so beside the
# production library; a process loads it with MBT_LIBRARY_VARIANT=name and the same MBT_EXTRA_HIPCC_FLAGS (they are part of the source hash).
if not VARIANT:
    VARIANT = os.environ.get("MBT_BUILD_VARIANT", "").strip()


def variant_path(variant: str) -> str:
    return LIB_PATH if not variant else os.path.join(PKG_DIR, f"libmbtenv.{variant}.so")


def sanitizer_runtime(variant: str) -> str:
    """The shared runtime to LD_PRELOAD for a sanitizer variant of the library (hipcc's own clang ships it)."""
    name = {"asan": "libclang_rt.asan-x86_64.so", "tsan": "libclang_rt.tsan-x86_64.so"}[variant]
    hits = glob.glob(os.path.join("/opt/rocm/lib/llvm/lib/clang", "*", "lib", "linux", name))
    if not hits:
        raise RuntimeError(f"{name} not found under /opt/rocm/lib/llvm")
    return sorted(hits)[-1]


def source_files():
    """Everything the library is compiled from, in a fixed order."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    return files + [os.path.normpath(HEADER)]


EMBEDDED = os.path.join(CSRC, "embedded_sources.inc")


def write_embedded_sources():
    """csrc/philox.hpp and csrc/step_kernel.hpp as C++ string constants: the source hiprtc compiles for user-defined
    plugins (mbt_env_create_jit) is the source of the ahead-of-time kernels, byte for byte.  A build artefact (ignored by git
    and by the source hash, which covers the two headers themselves)."""
    def literal(path):
        with open(path) as f:
            text = f.read()
        assert ')MBTSRC"' not in text
        # raw literals are limited to 64 KiB by some compilers: adjacent literals of at most 16 KiB each are concatenated
        parts, step = [], 16000
        for at in range(0, len(text), step):
            parts.append('R"MBTSRC(' + text[at:at + step] + ')MBTSRC"')
        return "\n".join(parts)

    body = ("// generated by mbt_gym_amd/build.py - do not edit\n"
            "namespace {\n"
            "const char* const kEmbeddedPhilox =\n" + literal(os.path.join(CSRC, "philox.hpp")) + ";\n"
            "const char* const kEmbeddedStepKernel =\n" + literal(os.path.join(CSRC, "step_kernel.hpp")) + ";\n"
            "}  // namespace\n")
    if not os.path.exists(EMBEDDED) or open(EMBEDDED).read() != body:
        with open(EMBEDDED, "w") as f:
            f.write(body)


def source_hash() -> str:
    """sha256 over every source of the library AND the extra compiler flags (MBT_EXTRA_HIPCC_FLAGS): a library built with
    experimental flags is stale for a process that runs without them, and the other way round."""
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode())
        h.update(b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    extra = os.environ.get("MBT_EXTRA_HIPCC_FLAGS", "").split()
    if extra:
        h.update(b"flags\0" + " ".join(extra).encode())
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


def _stale(path: str = LIB_PATH) -> bool:
    return library_hash(path) != source_hash()


def translation_units():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _object_key(unit: str, flags) -> str:
    """What one object file depends on: its own source, every header (each unit includes the kernel headers), the flags."""
    h = hashlib.sha256()
    for path in [unit] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.normpath(HEADER), EMBEDDED]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build_native(force: bool = False, verbose: bool = False, jobs: int = None) -> str:
    """Compile the HIP extension if it is missing or was built from different sources; returns the .so path.
    Safe to call from several processes at once (N ranks importing a stale tree): one holds the lock on the object directory and
    builds, the others wait and find the library current; temporary files carry the builder's pid."""
    target = variant_path(VARIANT)
    if not force and not _stale(target):
        return target
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmbtenv.so cannot be built (and there is no CPU fallback)")
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(target):  # another process built it while this one waited
                return target
            return _build_locked(hipcc, target, force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc, target, force, verbose, jobs):
    write_embedded_sources()
    digest = source_hash()
    mine = f".{os.getpid()}.tmp"

    def compile_unit(unit):
        name = os.path.splitext(os.path.basename(unit))[0]
        # only the C ABI unit bakes the hash of ALL sources in (mbt_source_hash): the kernel units keep their cached objects
        # when something outside their own inputs changes - and only it is instrumented in a sanitizer build
        flags = HIPCC_FLAGS + ([f'-DMBT_SOURCE_HASH="{digest}"'] + SANITIZE_FLAGS if name == "mbt_env" else [])
        obj = os.path.join(OBJ_DIR, f"{name}.{_object_key(unit, flags)[:16]}.o")
        if force or not os.path.exists(obj):
            if not VARIANT:  # (a sanitizer / experiment build leaves the production objects in the cache)
                for old in glob.glob(os.path.join(OBJ_DIR, f"{name}.*.o")):
                    os.remove(old)
            cmd = [hipcc] + flags + ["-c", unit, "-o", obj + mine]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(obj + mine, obj)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    units = translation_units()
    with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
        objects = list(pool.map(compile_unit, units))
    tmp = target + mine
    cmd = [hipcc, "--offload-arch=gfx950"] + objects + LINK_FLAGS + ([f"-fsanitize={SANITIZE}", "-shared-libsan"] if SANITIZE else []) + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, target)  # a process that has the old library mapped keeps its (unlinked) file
    return target


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
