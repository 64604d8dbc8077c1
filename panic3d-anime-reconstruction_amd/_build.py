"""In-tree build of libpanic3d_hip.so (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libpanic3d_hip.so")
SOURCES = ["p3d_kernels.hip", "p3d_synthesis.hip", "p3d_conv_plain.hip", "p3d_conv_up.hip", "p3d_conv_up4.hip", "p3d_fir.hip", "p3d_torgb.hip",
           "p3d_mcubes.hip", "p3d_paste.hip"]
HEADERS = ["p3d_math.hpp", "p3d_decode.hpp", "p3d_conv_common.hpp", "p3d_conv_stage.hpp", os.path.join("..", "..", "include", "panic3d_hip.h"),
           os.path.join("..", "..", "include", "p3d_numerics.h"), os.path.join("..", "..", "include", "p3d_mc_table.h")]
# -fno-slp-vectorize: v_pk_fma_f32 runs at ~0.4x the flop rate of v_fma_f32 on gfx950 (tools/ubench/valu_rates.hip).
# -ffp-contract=off: the arithmetic contract (include/p3d_numerics.h) names every fma explicitly.
# -pragma-unroll-threshold: k_render<96,...>'s twelve inverse-CDF batches must unroll completely (their results live in a
#   register array); the default 16 k-instruction cap refuses and the array lands in scratch memory.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
               "-mllvm", "-pragma-unroll-threshold=200000", "-fPIC", "-shared"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256[:16] over the kernel sources and headers: identifies WHICH kernels a profile was captured from (the GPU box
    has no .git; profiles/pmc_latest.json stores this and bench.py refuses a capture from other sources)."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


RENDER_UNIT = ["p3d_kernels.hip", "p3d_decode.hpp", "p3d_math.hpp", os.path.join("..", "..", "include", "panic3d_hip.h"),
               os.path.join("..", "..", "include", "p3d_numerics.h")]


def render_source_hash():
    """sha256[:16] over the translation unit of the renderer kernels (p3d_kernels.hip and everything it includes): the kernels
    bench.py times.  A PMC capture of k_render stays valid while THIS is unchanged — an edit of the synthesis / mesh / paste
    units compiles into other objects and cannot change the renderer's code."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(RENDER_UNIT):
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


SYNTHESIS_UNIT = ["p3d_synthesis.hip", "p3d_conv_plain.hip", "p3d_conv_up.hip", "p3d_conv_up4.hip", "p3d_fir.hip", "p3d_torgb.hip", "p3d_conv_common.hpp",
                  "p3d_conv_stage.hpp", os.path.join("..", "..", "include", "panic3d_hip.h")]


def synthesis_source_hash():
    """sha256[:16] over the translation unit of the convolution kernels (p3d_synthesis.hip and what it includes): the key of a
    recorded MFMA-utilisation capture (profiles/*_mfma_util.json), like render_source_hash() for the renderer's counters."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(SYNTHESIS_UNIT):
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def built_hash(so=SO):
    """The source hash a library was compiled from (embedded by build() as -DP3D_SRC_HASH in p3d_build_info's string), or None.
    Read from the FILE, not through dlopen: a library loaded here to ask it would stay mapped under its path, and the dlopen that
    follows a rebuild in the same process would get that stale image back (the loader matches loaded objects by name first)."""
    import re
    try:
        with open(so, "rb") as f:
            m = re.search(rb"libpanic3d_hip gfx950[^\0]{0,64}src=([0-9a-f]{16})", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def sources_present():
    return all(os.path.exists(os.path.join(CSRC, s)) for s in SOURCES + HEADERS)


def needs_build():
    """True when the library is missing or was compiled from other sources than the ones on disk (content hash, not mtimes:
    a copied tree keeps no useful timestamps).  A deployment that ships only the .so (no csrc/) has nothing to compare with:
    the existing library is trusted."""
    if not os.path.exists(SO):
        return True
    if not sources_present():
        return False
    return built_hash() != source_hash()


OBJ_DIR = os.path.join(os.path.dirname(HERE), "build", "obj")  # (git- and gpurun-ignored)
HASH_UNIT = "p3d_paste.hip"  # the translation unit that holds p3d_build_info(): the only one compiled with -DP3D_SRC_HASH


def _compile_objects(verbose=False):
    """One object per source, the sources compiled IN PARALLEL and cached by content (source + every header + flags): editing one
    .hip recompiles that file (and the few-second unit that embeds the source hash), not the 2.5-minute render kernels."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    hdr = hashlib.sha256()
    for hname in sorted(HEADERS):
        with open(os.path.join(CSRC, hname), "rb") as f:
            hdr.update(hname.encode() + b"\0" + f.read())
    jobs, objs = [], []
    for src in SOURCES:
        extra = [f'-DP3D_SRC_HASH="{source_hash()}"'] if src == HASH_UNIT else []
        k = hashlib.sha256(hdr.digest() + " ".join(flags + extra).encode())
        with open(os.path.join(CSRC, src), "rb") as f:
            k.update(f.read())
        obj = os.path.join(OBJ_DIR, f"{src}.{k.hexdigest()[:16]}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((src, [_hipcc()] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj + f".{os.getpid()}.tmp"], obj))

    def run(job):
        src, cmd, obj = job
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(cmd[-1], obj)
        finally:
            if os.path.exists(cmd[-1]):
                os.remove(cmd[-1])
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
        keep = set(objs)  # drop the objects of older source versions
        for fn in os.listdir(OBJ_DIR):
            if fn.endswith(".o") and os.path.join(OBJ_DIR, fn) not in keep:
                os.remove(os.path.join(OBJ_DIR, fn))
    return objs


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> libpanic3d_hip.so next to this file.  Returns the path.
    Safe under torch.distributed.run (every rank may get here at once): one process compiles under an exclusive file lock into
    its own temporary file and renames it into place; the others wait on the lock and then find the library up to date."""
    import fcntl
    if not force and not needs_build():
        return SO
    with open(SO + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another rank built it while this one waited
                return SO
            tmp = f"{SO}.{os.getpid()}.tmp"
            try:
                objs = _compile_objects(verbose)
                cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, SO)  # atomic: a process that is dlopen-ing the old file keeps its inode
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
