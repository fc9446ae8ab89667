"""Build libesmk.so (the gfx950 HIP engine) in-tree with hipcc.

    python -m esm_amd.build            # builds esm_amd/lib/libesmk.so unless it was built from these sources

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels
with the source tree to the GPU box.  The library carries the SHA-256 of the sources it was compiled from
(``esmk_version()`` -> "... src:<16 hex>"); ``needs_build`` compares that with the sources on disk, not mtimes,
and ``__graft_entry__.build`` / ``bench.py`` read it back through the C ABI.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libesmk.so")
SOURCES = ["gemm.hip", "gemm8.hip", "gemm9.hip", "gemm32.hip", "attention.hip", "attention128.hip", "elementwise.hip", "contacts.hip", "engine.hip", "engine_msa.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_epi.h", "engine_internal.h", os.path.join("..", "..", "include", "esmk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# build-time experiments (e.g. ESMK_HIPCC_EXTRA="-DESMK_G9_ALIGN=6"): part of the source hash, so such a library never
# passes for the default one
FLAGS += os.environ.get("ESMK_HIPCC_EXTRA", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libesmk.so cannot be built")


_MARK = b"esmk-src:"


def source_hash():
    """First 16 hex digits of the SHA-256 over every source and header (names and contents) and the flags."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_hash(path=LIB):
    """The source hash embedded in a built libesmk.so (read from the file, no dlopen), or None."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    return blob[i + len(_MARK): i + len(_MARK) + 16].decode("ascii", "replace") if i >= 0 else None


def needs_build():
    return library_hash() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    srchash = source_hash()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src == "engine.hip":
            cmd.insert(-4, f'-DESMK_SRC_HASH="{srchash}"')
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if library_hash() != srchash:
        raise RuntimeError("libesmk.so does not carry the hash of the sources it was just built from")
    # every kernel the host code launches must have been emitted (an uninstantiable template leaves an undefined
    # __device_stub__ symbol that only shows up at dlopen time)
    import ctypes

    ctypes.CDLL(LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
