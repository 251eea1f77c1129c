"""ctypes binding of libindelpost_hip.so (include/indelpost_hip.h).

The library is built in-tree by :func:`build` (hipcc, gfx950 only).  Loading fails loudly when the
shared object is missing; creating a context fails loudly when there is no GPU -- there is no CPU
fallback anywhere in this package.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libindelpost_hip.so")
CSRC = os.path.join(PKG_DIR, "csrc")
SRC = os.path.join(CSRC, "ipx_runtime.hip")
# the library is several translation units (compiled in parallel): the runtime + every other kernel, and twenty
# units holding the explicit instantiations of the striped-DP kernel families (csrc/ipx_kernels.h, end of file)
UNITS = [SRC] + [os.path.join(CSRC, "ipx_dp_%s.hip" % u) for u in "abcdefghijklmnopqvwxyz"]
HEADERS = [os.path.join(CSRC, h) for h in ("ipx_simt.h", "ipx_types.h", "ipx_kernels.h", "ipx_pipeline.h")] + [
    os.path.join(os.path.dirname(PKG_DIR), "include", "indelpost_hip.h")]
BUILD_DIR = os.path.join(CSRC, "build")

# numpy view of ipx_result (32 bytes)
RESULT_DTYPE = np.dtype([
    ("score1", "<u2"), ("score2", "<u2"), ("ref_begin1", "<i4"), ("ref_end1", "<i4"),
    ("read_begin1", "<i4"), ("read_end1", "<i4"), ("ref_end2", "<i4"), ("cigar_off", "<u4"),
    ("cigar_len", "<u2"), ("flag", "u1"), ("mode", "u1")])
assert RESULT_DTYPE.itemsize == 32

EXPORTS = [
    # reference-compatible four-call interface (ssw.h:86,91,126-134,139)
    "ssw_init", "init_destroy", "ssw_align", "align_destroy",
    # batched interface
    "ipx_device_count", "ipx_create", "ipx_destroy", "ipx_last_error", "ipx_set_params", "ipx_set_routing", "ipx_upload",
    "ipx_run", "ipx_sync", "ipx_download", "ipx_download_async", "ipx_wait", "ipx_set_async_io", "ipx_pin_host",
    "ipx_unpin_host", "ipx_align_batch", "ipx_set_profiling",
    "ipx_num_kernel_classes", "ipx_kernel_class_name", "ipx_kernel_times", "ipx_kernel_units", "ipx_last_run_ms", "ipx_debug_tb_counts", "ipx_debug_reruns",
    "ipx_synth_window", "ipx_synth_reads", "ipx_synth_mixed", "ipx_format_cigars", "ipx_cigar_hashes", "ipx_record_digest", "ipx_concat_sizes", "ipx_concat_tables", "ipx_group_by_length",
]


class IpxError(RuntimeError):
    pass


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in sources)


def needs_build():
    return _stale(LIB_PATH, UNITS + HEADERS)


def build(force=False, verbose=False, jobs=None, variant=None):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU): one object per translation unit,
    in parallel, then one link.  Safe under torchrun, where every rank may find the library stale at once: the build is
    serialised by a file lock, objects and the library are written under temporary names and renamed into place."""
    if variant is not None:
        return _build_variant(variant, force, verbose, jobs)
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB_PATH):
            return LIB_PATH          # prebuilt library travelled with the tree
        raise IpxError("hipcc not found and %s is not built" % LIB_PATH)
    os.makedirs(BUILD_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(BUILD_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():                    # another process built it while we waited
                return LIB_PATH
            return _build_locked(hipcc, force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


# TEST variants of the library (same sources, one more define; built next to the tests, never loaded by the package):
#   "noasm": -DIPX_NO_STRIPE_ASM -- the striped recurrence in its plain C++ form instead of the hand-scheduled inline asm (the form the
#            CPU emulator runs): tests/test_gpu_parity.py holds the two against each other on the GPU
VARIANTS = {"noasm": ["-DIPX_NO_STRIPE_ASM"]}


def variant_path(variant):
    return os.path.join(os.path.dirname(PKG_DIR), "tests", "variants", "libindelpost_hip_%s.so" % variant)


def _build_variant(variant, force, verbose, jobs):
    out = variant_path(variant)
    if not force and not _stale(out, UNITS + HEADERS):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(out):
            return out
        raise IpxError("hipcc not found and %s is not built" % out)
    bdir = os.path.join(CSRC, "build_" + variant)
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    return _build_locked(hipcc, True, verbose, jobs, bdir, out, VARIANTS[variant])


def _build_locked(hipcc, force, verbose, jobs, build_dir=None, lib_path=None, extra=()):
    BUILD_DIR_, LIB_PATH_ = build_dir or BUILD_DIR, lib_path or LIB_PATH
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + list(extra)
    tag = ".tmp%d" % os.getpid()
    objs, todo = [], []
    for src in UNITS:
        obj = os.path.join(BUILD_DIR_, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            todo.append((obj, [hipcc] + flags + ["-c", src, "-o", obj + tag]))
    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 2)))
    running, pending = [], list(todo)
    failed = None
    while pending or running:
        while pending and len(running) < jobs:
            obj, cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((obj, cmd, subprocess.Popen(cmd)))
        obj, cmd, p = running.pop(0)
        if p.wait() != 0:
            failed = failed or subprocess.CalledProcessError(p.returncode, cmd)
        else:
            os.replace(obj + tag, obj)
    if failed:
        raise failed
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH_ + tag] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(LIB_PATH_ + tag, LIB_PATH_)
    return LIB_PATH_


_lib = None


def lib():
    """Load the library (once) and declare the C signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IpxError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % LIB_PATH)
    _lib = load(LIB_PATH)
    return _lib


def load(path):
    """a shared object with the library's C ABI (the product library, or a test variant of it), signatures declared"""
    # HIP gives a process four hardware queues by default and multiplexes its streams onto them; two aligners in rotation (eight streams:
    # while one batch computes, the next one's slices copy in) then wait for each other inside a shared queue.  Ten queues -- the eight
    # streams and what the runtime uses itself: with exactly eight, every other list of the many-loci stream still took 26 ms instead of
    # 14 -- unless the caller has said otherwise; read by the HIP runtime when it initialises, i.e. at the first call into the library
    # (r04, many-loci stream through align_loci_stream: 4 queues 45.7, 8 queues 60-65, 10 queues 73-78 M aln/s; batches resident in HBM
    # within +-1 % at 4, 8, 10 or 16)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")
    L = C.CDLL(path)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.ipx_device_count.restype = C.c_int
    L.ipx_create.restype = vp
    L.ipx_create.argtypes = [C.c_int]
    L.ipx_destroy.argtypes = [vp]
    L.ipx_destroy.restype = None
    L.ipx_last_error.restype = C.c_char_p
    L.ipx_set_params.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ipx_set_routing.argtypes = [vp, C.c_int]
    L.ipx_set_routing.restype = C.c_int
    L.ipx_upload.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32]
    L.ipx_run.argtypes = [vp]
    L.ipx_sync.argtypes = [vp]
    L.ipx_download.argtypes = [vp, vp, vp, i64, C.POINTER(i64)]
    L.ipx_download_async.argtypes = [vp, vp, vp, i64, C.POINTER(i64)]
    L.ipx_wait.argtypes = [vp]
    L.ipx_set_async_io.argtypes = [vp, C.c_int]
    L.ipx_pin_host.argtypes = [vp, i64]
    L.ipx_unpin_host.argtypes = [vp]
    L.ipx_align_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, i64, C.POINTER(i64)]
    L.ipx_set_profiling.argtypes = [vp, C.c_int]
    L.ipx_num_kernel_classes.restype = C.c_int
    L.ipx_kernel_class_name.restype = C.c_char_p
    L.ipx_kernel_class_name.argtypes = [C.c_int]
    L.ipx_kernel_times.argtypes = [vp, vp, vp]
    L.ipx_kernel_units.argtypes = [vp, vp]
    L.ipx_kernel_units.restype = C.c_int
    L.ipx_last_run_ms.restype = C.c_float
    L.ipx_last_run_ms.argtypes = [vp]
    L.ipx_debug_tb_counts.argtypes = [vp, vp]
    L.ipx_debug_reruns.argtypes = [vp]
    L.ipx_debug_reruns.restype = C.c_int
    L.ipx_debug_tb_counts.restype = C.c_int
    L.ipx_synth_window.restype = C.c_uint64
    L.ipx_synth_window.argtypes = [C.c_uint64, vp, i32]
    L.ipx_format_cigars.restype = C.c_int64
    L.ipx_format_cigars.argtypes = [vp, vp, i64, vp, i64, vp]
    L.ipx_cigar_hashes.restype = None
    L.ipx_cigar_hashes.argtypes = [vp, vp, i64, vp]
    L.ipx_record_digest.restype = C.c_uint64
    L.ipx_record_digest.argtypes = [vp, vp, i64]
    L.ipx_concat_sizes.restype = C.c_int
    L.ipx_concat_sizes.argtypes = [vp, i64, vp]
    L.ipx_concat_tables.restype = C.c_int
    L.ipx_concat_tables.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ipx_group_by_length.restype = C.c_int
    L.ipx_group_by_length.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.ipx_synth_reads.restype = C.c_uint64
    L.ipx_synth_reads.argtypes = [C.c_uint64, vp, i32, vp, i64, i32]
    L.ipx_synth_mixed.restype = i64
    L.ipx_synth_mixed.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp]
    for f in ("ipx_set_params", "ipx_set_routing", "ipx_upload", "ipx_run", "ipx_sync", "ipx_download", "ipx_download_async", "ipx_wait", "ipx_set_async_io", "ipx_pin_host",
    "ipx_unpin_host", "ipx_align_batch",
              "ipx_set_profiling", "ipx_kernel_times"):
        getattr(L, f).restype = C.c_int
    return L


def last_error():
    return lib().ipx_last_error().decode(errors="replace")
