"""Batched striped Smith-Waterman on MI355X: job tables, GPU contexts, multi-GPU sharding.

This is the batched counterpart of the reference's per-read loop
``setRead -> ssw_init -> ssw_align -> align_destroy`` (indelpost/localn.py:464-472,
indelpost/sswpy.pyx:149-178, 199-225): a *job* is (read, window id, gap_open, gap_extension); a
batch of jobs is aligned by one call into libindelpost_hip.so.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import RESULT_DTYPE, IpxError

_OPS = "MIDNSHP=X"

# speed-only routing switches of the library (ipx_set_routing; csrc/ipx_types.h IPX_ROUTE_*): every combination gives
# identical results -- the tests run them all
ROUTE_NO_WORD_FIRST, ROUTE_NO_PERM_PROFILE, ROUTE_NO_BRACKET, ROUTE_TB_NO_FUSE, ROUTE_NO_MC_LDS, ROUTE_NO_F16, ROUTE_NO_SKEW, ROUTE_NO_VL2 = 1, 2, 4, 8, 16, 32, 64, 128
ROUTE_NO_PLAIN_FIRST, ROUTE_NO_CLASS_MERGE, ROUTE_NO_TIERS, ROUTE_NO_EXACT_DIRECT, ROUTE_TB_NO_WAVE_PER_JOB = 256, 512, 1024, 2048, 4096
ROUTE_TB_NO_UNGAPPED, ROUTE_TB_NO_DIAG, ROUTE_NO_LAT, ROUTE_FORCE_LAT, ROUTE_NO_LAT_PROOF, ROUTE_NO_SPECULATE, ROUTE_TEST_SKIP_ALL = 8192, 16384, 32768, 65536, 131072, 524288, 2097152
ROUTE_NO_WIDE, ROUTE_NO_SETPRIO, ROUTE_TB_PER_WIDTH, ROUTE_REV_BELOW, ROUTE_NO_BAND_REV, ROUTE_FORCE_BAND_REV = 262144, 4194304, 8388608, 16777216, 33554432, 67108864

# DNA_BASE_LUT of the reference (sswpy.pyx:16-25): A/a 0, C/c 1, G/g 2, T/t 3, U/u 0, else 4.
# Bytes >= 128 index the reference's table out of bounds (undefined); they map to N here.
DNA_LUT = np.full(256, 4, np.int8)
for _c, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 0)):
    DNA_LUT[ord(_c)] = _v
    DNA_LUT[ord(_c.lower())] = _v


def encode_dna(seq):
    """str/bytes -> int8 codes (dnaToInt8, sswpy.pyx:27-29)."""
    if isinstance(seq, str):
        seq = seq.encode("utf8")
    return DNA_LUT[np.frombuffer(seq, np.uint8)]


def dna_score_matrix(match_score, mismatch_penalty):
    """5x5 int8 matrix of SSW.buildDNAScoreMatrix (sswpy.pyx:306-336): both arguments pass through
    uint8 (sswpy.pyx:128-129) and are then stored as int8; the N row and column are 0."""
    ms = np.array([int(match_score) & 255], np.uint8).astype(np.int8)[0]
    mm = np.array([(-(int(mismatch_penalty) & 255)) & 255], np.uint8).astype(np.int8)[0]
    m = np.zeros((5, 5), np.int8)
    m[:4, :4] = mm
    for i in range(4):
        m[i, i] = ms
    return m.reshape(-1)


def cigar_to_string(ops):
    """BAM-encoded uint32 ops -> '12M1D...' exactly as sswpy.pyx:283-289."""
    return "".join("%d%s" % (int(c) >> 4, _OPS[int(c) & 15] if (int(c) & 15) <= 8 else "M") for c in ops)


class JobTable:
    """Concatenated reads/windows plus per-job window id and gap penalties (host, numpy)."""

    def __init__(self, reads, read_off, refs, ref_off, ref_id, gap_open, gap_ext, mask_len=None):
        self.reads = np.ascontiguousarray(reads, np.int8)
        self.read_off = np.ascontiguousarray(read_off, np.int64)
        self.refs = np.ascontiguousarray(refs, np.int8)
        self.ref_off = np.ascontiguousarray(ref_off, np.int64)
        self.ref_id = np.ascontiguousarray(ref_id, np.int32)
        n = len(self.ref_id)
        self.gap_open, self.gap_ext = self._narrow(gap_open, n), self._narrow(gap_ext, n)
        self.mask_len = None if mask_len is None else np.ascontiguousarray(mask_len, np.int32)
        if len(self.read_off) != n + 1:
            raise ValueError("read_off must have n_jobs+1 entries")

    @staticmethod
    def _narrow(g, n):
        """Python ints narrow to uint8 at the C boundary (ssw.h:129-130); arrays that already are uint8 pass through"""
        if isinstance(g, np.ndarray) and g.dtype == np.uint8 and g.shape == (n,) and g.flags.c_contiguous:
            return g
        return np.ascontiguousarray(np.broadcast_to(np.asarray(g, np.int64) & 255, (n,)), np.uint8)

    @property
    def n_jobs(self):
        return len(self.ref_id)

    @property
    def n_refs(self):
        return len(self.ref_off) - 1

    @classmethod
    def from_sequences(cls, reads, refs, ref_id, gap_open, gap_ext, encoded=False):
        """reads / refs: lists of str/bytes (or int8 arrays when encoded=True)."""
        enc = (lambda s: np.asarray(s, np.int8)) if encoded else encode_dna
        r = [enc(s) for s in reads]
        f = [enc(s) for s in refs]
        ro = np.zeros(len(r) + 1, np.int64)
        fo = np.zeros(len(f) + 1, np.int64)
        if r:
            ro[1:] = np.cumsum([len(x) for x in r])
        if f:
            fo[1:] = np.cumsum([len(x) for x in f])
        rc = np.concatenate(r) if r and ro[-1] else np.zeros(0, np.int8)
        fc = np.concatenate(f) if f and fo[-1] else np.zeros(0, np.int8)
        return cls(rc, ro, fc, fo, ref_id, gap_open, gap_ext)

    def desc(self):
        """80 bytes = ten little-endian int64: addresses of reads, read_off, refs, ref_off, ref_id, gap_open, gap_ext, mask_len (or 0), then
        n_jobs, n_refs -- what the library's ipx_concat_tables takes per table.  Cached (the arrays are owned by the table and never
        reallocated); as BYTES, so that the descriptors of ten thousand tables are one b"".join away from the array the library reads."""
        d = getattr(self, "_descb", None)
        if d is None:
            import struct
            d = self._descb = struct.pack("<10q", self.reads.ctypes.data, self.read_off.ctypes.data, self.refs.ctypes.data, self.ref_off.ctypes.data,
                                          self.ref_id.ctypes.data, self.gap_open.ctypes.data, self.gap_ext.ctypes.data,
                                          0 if self.mask_len is None else self.mask_len.ctypes.data, len(self.ref_id), len(self.ref_off) - 1)
        return d

    @classmethod
    def concat(cls, tables, staging=None):
        """Many job tables as ONE (the many-loci entry: a locus' few hundred jobs are far too few to fill a GPU, a thousand loci are
        not): jobs in table order, windows renumbered.  r04: the copying is the library's (ipx_concat_tables: memcpy per array and
        table, several threads); Python only gathers ten integers per table (JobTable.desc, cached on the table) -- r03's pure-numpy
        form spent 39 ms on the list handling of 12 500 tables.  mask_len: kept when every table has it, else the default rule.
        staging: a callable (read_bytes, window_bytes, jobs, windows, with_mask) -> the eight arrays to fill (reads, read_off, refs,
        ref_off, ref_id, gap_open, gap_ext, mask_len or None), e.g. an aligner's page-locked, reused buffers
        (MultiStreamAligner.loci_staging); default: fresh arrays."""
        import operator
        tables = tables if isinstance(tables, (list, tuple)) else list(tables)
        n = len(tables)
        if not n:
            return cls(np.zeros(0, np.int8), np.zeros(1, np.int64), np.zeros(0, np.int8), np.zeros(1, np.int64), np.zeros(0, np.int32), 0, 0)
        L = _lib.lib()
        try:                                                       # (every table has been here before: one attribute read per table)
            blob = b"".join(map(operator.attrgetter("_descb"), tables))
        except (AttributeError, TypeError):
            blob = b"".join([t.desc() for t in tables])
        desc = np.frombuffer(blob, np.int64)
        tot = np.zeros(4, np.int64)
        all_mask = L.ipx_concat_sizes(desc.ctypes.data, n, tot.ctypes.data)
        rb, fb, nj, nr = (int(x) for x in tot)
        if staging is not None:
            reads, read_off, refs, ref_off, ref_id, go, ge, mask = staging(rb, fb, nj, nr, bool(all_mask))
        else:
            reads, read_off = np.empty(rb, np.int8), np.empty(nj + 1, np.int64)
            refs, ref_off = np.empty(fb, np.int8), np.empty(nr + 1, np.int64)
            ref_id, go, ge = np.empty(nj, np.int32), np.empty(nj, np.uint8), np.empty(nj, np.uint8)
            mask = np.empty(nj, np.int32) if all_mask else None
        rc = L.ipx_concat_tables(desc.ctypes.data, n, reads.ctypes.data, read_off.ctypes.data, refs.ctypes.data, ref_off.ctypes.data,
                                 ref_id.ctypes.data, go.ctypes.data, ge.ctypes.data, None if mask is None else mask.ctypes.data)
        if rc != 0:
            raise IpxError("ipx_concat_tables: %s" % L.ipx_last_error().decode())
        out = cls(reads, read_off, refs, ref_off, ref_id, go, ge, mask)
        out.table_jobs = desc[8::10].copy()                         # jobs per input table (BatchResult.split takes it)
        return out

    def grouped_by_length(self):
        """(table, order): the same jobs GROUPED BY READ LENGTH (stable), order[k] = the job of this table that comes k-th.  The library's
        copy (ipx_group_by_length: counting sort + one memcpy per job, several threads; ~10 ms per million jobs).  Windows are shared, not copied."""
        L = _lib.lib()
        n = self.n_jobs
        order = np.empty(n, np.uint32)
        reads, read_off = np.empty(len(self.reads), np.int8), np.empty(n + 1, np.int64)
        ref_id, go, ge = np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(n, np.uint8)
        mask = None if self.mask_len is None else np.empty(n, np.int32)
        src = [np.ascontiguousarray(a) for a in (self.reads, self.read_off, self.ref_id, self.gap_open, self.gap_ext)]
        rc = L.ipx_group_by_length(src[0].ctypes.data, src[1].ctypes.data, src[2].ctypes.data, src[3].ctypes.data, src[4].ctypes.data,
                                   None if mask is None else np.ascontiguousarray(self.mask_len).ctypes.data, n, order.ctypes.data, reads.ctypes.data,
                                   read_off.ctypes.data, ref_id.ctypes.data, go.ctypes.data, ge.ctypes.data, None if mask is None else mask.ctypes.data)
        if rc != 0:
            raise IpxError("ipx_group_by_length: %s" % L.ipx_last_error().decode())
        return JobTable(reads, read_off, self.refs, self.ref_off, ref_id, go, ge, mask), order

    def shard(self, lo, hi):
        """Contiguous job range [lo, hi) with only the windows it references."""
        rid = self.ref_id[lo:hi]
        base = self.read_off[lo]
        if len(self.refs) <= (1 << 20):       # few / small windows: ship them all, no renumbering (views, no copies)
            return JobTable(self.reads[base:self.read_off[hi]], self.read_off[lo:hi + 1] - base, self.refs, self.ref_off,
                            rid, self.gap_open[lo:hi], self.gap_ext[lo:hi],
                            None if self.mask_len is None else self.mask_len[lo:hi])
        mask = None if self.mask_len is None else self.mask_len[lo:hi]
        reads, read_off = self.reads[base:self.read_off[hi]], self.read_off[lo:hi + 1] - base
        if len(rid) == 0:
            return JobTable(reads, read_off, np.zeros(0, np.int8), np.zeros(1, np.int64), rid, self.gap_open[lo:hi],
                            self.gap_ext[lo:hi], mask)
        r0, r1 = int(rid.min()), int(rid.max())
        span = int(self.ref_off[r1 + 1] - self.ref_off[r0])
        if span <= 2 * int(self.read_off[hi] - base) + (1 << 20):
            # the shard's windows form a (nearly) dense id range, as in a table sorted by locus: ship that range as it
            # lies (views), renumbering is one subtraction
            return JobTable(reads, read_off, self.refs[self.ref_off[r0]:self.ref_off[r1 + 1]],
                            self.ref_off[r0:r1 + 2] - self.ref_off[r0], rid - np.int32(r0), self.gap_open[lo:hi],
                            self.gap_ext[lo:hi], mask)
        used, inv = np.unique(rid, return_inverse=True)          # scattered windows: gather them with one fancy index
        starts = self.ref_off[used]
        lens = self.ref_off[used + 1] - starts
        fo = np.zeros(len(used) + 1, np.int64)
        np.cumsum(lens, out=fo[1:])
        idx = np.repeat(starts - fo[:-1], lens) + np.arange(int(fo[-1]), dtype=np.int64)
        return JobTable(reads, read_off, self.refs[idx], fo, inv.astype(np.int32), self.gap_open[lo:hi],
                        self.gap_ext[lo:hi], mask)


def record_digest(rec, cigar_hash):
    """64-bit digest (XXH64, seed 0) of a batch's results in job order: every public field of every record -- score1, score2, the five
    coordinates, flag, cigar_len -- and, per job, the FNV-1a hash of its BAM-encoded CIGAR ops, ten little-endian int64 per job.
    bench.py compares it with the digest of the reference's results on the same job table (tests/golden/bench_digests.json,
    oracle/gen_bench_digests.py).  Computed by the library's host-side helper (ipx_record_digest): no third-party hashing package is
    needed (tests/test_host_logic.py holds it against the `xxhash` package where that is installed).
    (r02 used a position-weighted sum; a real hash cannot cancel.)"""
    from . import _lib
    if rec.dtype != RESULT_DTYPE:                # (the CPU checker's records: same fields in another layout)
        r2 = np.zeros(len(rec), RESULT_DTYPE)
        for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "flag", "cigar_len"):
            r2[f] = rec[f]
        rec = r2
    rec = np.ascontiguousarray(rec)
    ch = np.ascontiguousarray(cigar_hash, np.uint32)
    assert len(ch) == len(rec)
    return int(_lib.lib().ipx_record_digest(rec.ctypes.data, ch.ctypes.data, len(rec)))


class _SplitResults:
    """list-like: the per-table BatchResults of BatchResult.split"""

    def __init__(self, whole, counts):
        self._whole = whole
        c = np.asarray(counts, np.int64)
        self._off = np.zeros(len(c) + 1, np.int64)
        np.cumsum(c, out=self._off[1:])

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return BatchResult(self._whole.records[int(self._off[k]):int(self._off[k + 1])], self._whole.cigar_pool)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class BatchResult:
    """Result records (numpy structured array, RESULT_DTYPE) + the cigar pool."""

    def __init__(self, records, cigar_pool):
        self.records = records
        self.cigar_pool = cigar_pool

    def __len__(self):
        return len(self.records)

    def copy(self):
        """a result that owns its arrays (results collected into pinned buffers are views that a later collect() overwrites)"""
        return BatchResult(self.records.copy(), np.array(self.cigar_pool, np.uint32, copy=True))

    def split(self, counts):
        """the results of a concatenated job table (JobTable.concat) back as one BatchResult per table: record views that share
        the cigar pool (cigar_off stays an index into it).  A sequence that makes the views when they are asked for (r04: building
        12 500 result objects up front was a tenth of a many-loci call)."""
        return _SplitResults(self, counts)

    def cigar_ops(self, i):
        r = self.records[i]
        if r["cigar_len"] == 0:
            return None
        return self.cigar_pool[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])]

    def cigar_string(self, i):
        ops = self.cigar_ops(i)
        return None if ops is None else cigar_to_string(ops)

    def cigar_strings(self):
        """All CIGAR strings of the batch (None where the reference returns no cigar), formatted by the
        library in one call (sswpy.pyx:283-289) instead of one Python loop per alignment."""
        from . import _lib
        n = len(self.records)
        if n == 0:
            return []
        L = _lib.lib()
        rec = np.ascontiguousarray(self.records)
        pool = np.ascontiguousarray(self.cigar_pool, np.uint32)
        off = np.zeros(n + 1, np.int64)
        cap = 12 * int(rec["cigar_len"].astype(np.int64).sum()) + 16
        buf = np.zeros(cap, np.uint8)
        tot = L.ipx_format_cigars(rec.ctypes.data, pool.ctypes.data if pool.size else None, n, buf.ctypes.data, cap, off.ctypes.data)
        if tot < 0:
            raise RuntimeError("ipx_format_cigars: buffer too small")
        text = buf[:tot].tobytes().decode("ascii")
        o = off.tolist()
        has = (rec["cigar_len"] > 0).tolist()
        return [text[o[i]:o[i + 1]] if has[i] else None for i in range(n)]

    def cigar_wsums(self):
        """Per job: sum_q cigar[q]*(q+1) mod 2^32 (0 without a CIGAR), vectorised over the pool."""
        rec = self.records
        n = len(rec)
        cl = rec["cigar_len"].astype(np.int64)
        tot = int(cl.sum())
        if tot == 0:
            return np.zeros(n, np.uint32)
        idx = np.repeat(np.arange(n), cl)
        within = np.arange(tot) - np.repeat(np.cumsum(cl) - cl, cl)
        ops = self.cigar_pool.astype(np.int64)[np.repeat(rec["cigar_off"].astype(np.int64), cl) + within]
        # ops < 2^32 and within < 2^16: products and per-job sums stay far below 2^53, exact in float64
        return (np.bincount(idx, weights=(ops * (within + 1)).astype(np.float64), minlength=n).astype(np.uint64)
                & np.uint64(0xFFFFFFFF)).astype(np.uint32)

    def cigar_hashes(self):
        """per job: FNV-1a (32 bit) of its BAM-encoded CIGAR ops, 2166136261 without a CIGAR (one call into the library)"""
        from . import _lib
        n = len(self.records)
        out = np.zeros(n, np.uint32)
        if n:
            rec = np.ascontiguousarray(self.records)
            pool = np.ascontiguousarray(self.cigar_pool, np.uint32)
            _lib.lib().ipx_cigar_hashes(rec.ctypes.data, pool.ctypes.data if pool.size else None, n, out.ctypes.data)
        return out

    def digest(self):
        return record_digest(self.records, self.cigar_hashes())

    def as_dict(self, i):
        """Same keys as oracle.Backend.align() for direct comparison in tests."""
        r = self.records[i]
        ops = self.cigar_ops(i)
        return dict(score1=int(r["score1"]), score2=int(r["score2"]), ref_begin1=int(r["ref_begin1"]),
                    ref_end1=int(r["ref_end1"]), read_begin1=int(r["read_begin1"]),
                    read_end1=int(r["read_end1"]), ref_end2=int(r["ref_end2"]), flag=int(r["flag"]),
                    cigar=None if ops is None else [int(x) for x in ops])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class GpuAligner:
    """One GPU: HIP stream + HBM-resident batch + workspace (ipx_ctx)."""

    def __init__(self, device=0, match_score=2, mismatch_penalty=2, matrix=None, library=None):
        L = library if library is not None else _lib.lib()         # (library: a test variant of the shared object, _lib.load)
        self._L = L
        self._ctx = L.ipx_create(int(device))
        if not self._ctx:
            raise IpxError("ipx_create(%d) failed: %s" % (device, _lib.last_error()))
        self.device = device
        self._n_jobs = 0
        self.set_scoring(match_score, mismatch_penalty, matrix)

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.ipx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise IpxError("%s failed (%d): %s" % (what, rc, self._L.ipx_last_error().decode(errors="replace")))

    def set_scoring(self, match_score=2, mismatch_penalty=2, matrix=None, flag=1, filters=0, filterd=0,
                    score_size=2):
        m = dna_score_matrix(match_score, mismatch_penalty) if matrix is None else np.ascontiguousarray(matrix, np.int8)
        if m.size != 25:
            raise ValueError("matrix must be 5x5")
        self.matrix = m
        self._check(self._L.ipx_set_params(self._ctx, _p(m), flag, filters, filterd, score_size), "ipx_set_params")

    def set_routing(self, flags):
        """speed-only routing switches (ROUTE_*): which proofs / kernel variants are tried; never changes a result"""
        self._check(self._L.ipx_set_routing(self._ctx, int(flags)), "ipx_set_routing")

    # -- staged interface (bench.py: inputs resident in HBM before the timed region) --
    def upload(self, jobs):
        # The previous table stays referenced until ipx_upload returns: in asynchronous mode its H2D copies may still be in flight
        # and the library waits for them first thing; only then may its arrays (and the temporaries JobTable.shard made) go.
        self._check(self._L.ipx_upload(self._ctx, _p(jobs.reads), _p(jobs.read_off), _p(jobs.refs), _p(jobs.ref_off),
                                       _p(jobs.ref_id), _p(jobs.gap_open), _p(jobs.gap_ext), _p(jobs.mask_len),
                                       jobs.n_jobs, jobs.n_refs), "ipx_upload")
        self._jobs = jobs   # keep host arrays alive
        self._n_jobs = jobs.n_jobs

    def run(self):
        self._check(self._L.ipx_run(self._ctx), "ipx_run")

    def sync(self):
        self._check(self._L.ipx_sync(self._ctx), "ipx_sync")

    def download_into(self, rec, pool):
        """Download into caller-owned buffers (contiguous views are fine).  Returns the number of CIGAR ops written,
        or -(needed) when `pool` is too small (nothing usable was written then)."""
        used = C.c_int64(0)
        rc = self._L.ipx_download(self._ctx, C.c_void_p(rec.ctypes.data), C.c_void_p(pool.ctypes.data), len(pool), C.byref(used))
        if rc == -5 and used.value > len(pool):
            return -int(used.value)
        self._check(rc, "ipx_download")
        return int(used.value)

    def set_async_io(self, on=True):
        """promise that the arrays of an uploaded JobTable stay alive and unchanged until the next sync(), so upload() does
        not wait for its copies (the aligners keep a reference to the table they were given)"""
        self._check(self._L.ipx_set_async_io(self._ctx, int(bool(on))), "ipx_set_async_io")

    def download_async_into(self, rec, pool):
        """after sync(): start the copy of the records and CIGAR ops into caller-owned buffers; wait() completes it.
        Returns the number of CIGAR ops, or -(needed) when `pool` is too small (nothing is copied then)."""
        used = C.c_int64(0)
        rc = self._L.ipx_download_async(self._ctx, C.c_void_p(rec.ctypes.data), C.c_void_p(pool.ctypes.data), len(pool), C.byref(used))
        if rc == -5 and used.value > len(pool):
            return -int(used.value)
        self._check(rc, "ipx_download_async")
        return int(used.value)

    def wait(self):
        self._check(self._L.ipx_wait(self._ctx), "ipx_wait")

    def download(self, cigar_ops_per_job=16):
        n = self._n_jobs
        rec = np.zeros(n, RESULT_DTYPE)
        cap = max(1024, n * cigar_ops_per_job)
        while True:
            pool = np.empty(cap, np.uint32)
            used = C.c_int64(0)
            rc = self._L.ipx_download(self._ctx, _p(rec), _p(pool), cap, C.byref(used))
            if rc == -5 and used.value > cap:      # IPX_ERR_CIGAR_POOL: host pool too small
                cap = int(used.value) + 16
                continue
            self._check(rc, "ipx_download")
            return BatchResult(rec, pool[:used.value])

    def align(self, jobs):
        """upload + run + sync + download."""
        self.upload(jobs)
        for _ in range(6):
            self.run()
            rc = self._L.ipx_sync(self._ctx)
            if rc == -5:                           # device cigar pool exhausted: let the C side grow it
                return self._align_one_call(jobs)
            self._check(rc, "ipx_sync")
            break
        return self.download()

    def _align_one_call(self, jobs):
        n = jobs.n_jobs
        rec = np.zeros(n, RESULT_DTYPE)
        cap = max(1024, n * 64)
        while True:
            pool = np.zeros(cap, np.uint32)
            used = C.c_int64(0)
            rc = self._L.ipx_align_batch(self._ctx, _p(jobs.reads), _p(jobs.read_off), _p(jobs.refs), _p(jobs.ref_off),
                                         _p(jobs.ref_id), _p(jobs.gap_open), _p(jobs.gap_ext), _p(jobs.mask_len),
                                         n, jobs.n_refs, _p(rec), _p(pool), cap, C.byref(used))
            if rc == -5 and used.value > cap:
                cap = int(used.value) + 16
                continue
            self._check(rc, "ipx_align_batch")
            return BatchResult(rec, pool[:used.value])

    # -- measurement --
    def set_profiling(self, on):
        # on: False/0 off, True/1 events around every launch, 2 only around the striped DP kernels
        self._check(self._L.ipx_set_profiling(self._ctx, int(on)), "ipx_set_profiling")

    def kernel_times(self):
        k = self._L.ipx_num_kernel_classes()
        ms = np.zeros(k, np.float32)
        cnt = np.zeros(k, np.int32)
        self._check(self._L.ipx_kernel_times(self._ctx, _p(ms), _p(cnt)), "ipx_kernel_times")
        out = {}
        for i in range(k):
            if cnt[i]:
                name = self._L.ipx_kernel_class_name(i).decode()
                t, c = out.get(name, (0.0, 0))
                out[name] = (t + float(ms[i]), c + int(cnt[i]))
        return out

    def kernel_units(self):
        """{kernel class name: alignments its launches processed since set_profiling} (striped DP kernels only)."""
        k = self._L.ipx_num_kernel_classes()
        un = np.zeros(k, np.int64)
        self._check(self._L.ipx_kernel_units(self._ctx, _p(un)), "ipx_kernel_units")
        out = {}
        for i in np.flatnonzero(un):
            name = self._L.ipx_kernel_class_name(int(i)).decode()
            out[name] = out.get(name, 0) + int(un[i])
        return out

    def traceback_routing(self):
        """jobs of the last run per traceback list: [0..6] lane-per-job kernels, first band 1..7 (doubled bands included); [7] the
        wave-per-job kernel (k_tb_coop); [8..10] the anti-diagonal tiers of 16 / 32 / 64 lanes per job (k_tb_diag); [11] 0"""
        out = np.zeros(12, np.uint32)
        self._check(self._L.ipx_debug_tb_counts(self._ctx, _p(out)), "ipx_debug_tb_counts")
        return out.tolist()

    def reruns(self):
        """runs repeated because a pass predicted empty (latency tier: not launched) held a job after all"""
        return int(self._L.ipx_debug_reruns(self._ctx))

    def last_run_ms(self):
        return float(self._L.ipx_last_run_ms(self._ctx))


class LociStaging:
    """Page-locked, reused INPUT arrays for concatenated job tables: call it with the sizes JobTable.concat asks for and get the eight
    arrays to fill (reads, read_off, refs, ref_off, ref_id, gap_open, gap_ext, mask_len or None), cut to size.  The buffers grow to
    the largest batch seen (+25 %) and are registered with the HIP runtime once (hipHostRegister through ipx_pin_host), so that the copies
    in are asynchronous.  What was handed out stays valid until the next call.  Plain arrays on back-ends without the capability."""

    def __init__(self, library):
        self._L = library if library is not None and hasattr(library, "ipx_pin_host") else None
        self._st, self._pinned = None, []

    def __call__(self, rb, fb, nj, nr, with_mask):
        if self._L is None:
            return (np.empty(rb, np.int8), np.empty(nj + 1, np.int64), np.empty(fb, np.int8), np.empty(nr + 1, np.int64), np.empty(nj, np.int32),
                    np.empty(nj, np.uint8), np.empty(nj, np.uint8), np.empty(nj, np.int32) if with_mask else None)
        st, need = self._st, (rb, fb, nj + 1, nr + 1)
        if st is None or any(n > c for n, c in zip(need, st["cap"])) or (with_mask and st["mask"] is None):
            self.close()
            cap = tuple(int(n * 1.25) + 1024 for n in need)
            st = self._st = {"cap": cap, "reads": np.empty(cap[0], np.int8), "read_off": np.empty(cap[2], np.int64), "refs": np.empty(cap[1], np.int8),
                             "ref_off": np.empty(cap[3], np.int64), "ref_id": np.empty(cap[2], np.int32), "go": np.empty(cap[2], np.uint8),
                             "ge": np.empty(cap[2], np.uint8), "mask": np.empty(cap[2], np.int32) if with_mask else None}
            for a in (st["reads"], st["read_off"], st["refs"], st["ref_off"], st["ref_id"], st["go"], st["ge"], st["mask"]):
                if a is not None and a.nbytes and self._L.ipx_pin_host(C.c_void_p(a.ctypes.data), a.nbytes) == 0:
                    self._pinned.append(a)
        return (st["reads"][:rb], st["read_off"][:nj + 1], st["refs"][:fb], st["ref_off"][:nr + 1], st["ref_id"][:nj], st["go"][:nj], st["ge"][:nj],
                st["mask"][:nj] if with_mask else None)

    def close(self):
        for a in self._pinned:
            self._L.ipx_unpin_host(C.c_void_p(a.ctypes.data))
        self._st, self._pinned = None, []


class MultiStreamAligner:
    """Several HIP streams on ONE GPU: the job table is cut into `streams` contiguous slices, each
    slice runs the whole pipeline on its own stream (its own ipx_ctx and workspace).  The slices are
    independent, so the kernels of one slice fill the tails and the low-occupancy phases of the others:
    measured +30 % alignments/s at 3-4 streams on MI355X (1.2 M reads).  Same interface as GpuAligner."""

    def __init__(self, device=0, match_score=2, mismatch_penalty=2, matrix=None, streams=4, aligner_cls=None):
        self.device = device
        cls = GpuAligner if aligner_cls is None else aligner_cls
        self.parts = [cls(device, match_score, mismatch_penalty, matrix) for _ in range(max(1, streams))]
        self.min_jobs_per_stream = 50000
        self.balance_by_cells = False   # cut the stream slices by work (read length x window length) instead of by job count
        self._active = self.parts
        self._pinned = []            # host arrays page-locked by pin_host (kept alive here)
        self._out = None             # pinned (records, cigar pool) pairs, used in alternation by collect(): see pin_host
        self._out_turn = 0
        self._staging = None         # page-locked input staging of loci_staging (a LociStaging)
        self.group_by_length = "auto"   # upload(): group a big batch of mixed read lengths by length before cutting it (True / False / "auto")
        self._order = None           # ... the order that takes its results home (download())
        self._out_pinned, self._out_loci = [], False     # the output pairs ensure_outputs made

    def close(self):
        self.unpin()
        for p in self.parts:
            p.close()

    def __del__(self):                 # (an aligner dropped without close() must not leave page-locked arrays behind)
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- page-locked host buffers: transfers that overlap the other slices' kernels --
    def pin_host(self, jobs, cigar_ops_per_job=16):
        """Page-lock the arrays of `jobs` and a reusable pair of output buffers (hipHostRegister) and switch the contexts to
        asynchronous transfers.  Worth it when the same host buffers are used for many batches (registering 150 MB costs
        about as much as copying it once).  Returns False on back-ends without the capability (tests/emu).
        LIFETIME: with pinned buffers collect() / align() return views of one of TWO pinned output pairs used in turn: a result
        stays valid while the next batch is collected and is overwritten by the collect() after that; BatchResult.copy() detaches it."""
        L = getattr(self.parts[0], "_L", None)
        if L is None or not hasattr(L, "ipx_pin_host"):
            return False
        self.unpin()
        n = jobs.n_jobs
        caps = sum(max(1024, (b1 - b0) * cigar_ops_per_job) for b0, b1 in zip(shard_bounds(n, len(self.parts))[:-1], shard_bounds(n, len(self.parts))[1:]))
        # TWO output pairs used in alternation: the BatchResult of one collect() stays valid while the next batch is collected
        # (it is overwritten by the collect() after that; callers that keep results longer copy them: BatchResult.copy()).
        out = [(np.empty(n, RESULT_DTYPE), np.empty(caps, np.uint32)) for _ in (0, 1)]
        for a in (jobs.reads, jobs.read_off, jobs.refs, jobs.ref_off, jobs.ref_id, jobs.gap_open, jobs.gap_ext, jobs.mask_len) + out[0] + out[1]:
            if a is not None and a.nbytes and L.ipx_pin_host(C.c_void_p(a.ctypes.data), a.nbytes) == 0:
                self._pinned.append(a)
        self._out, self._out_loci = out, False
        for p in self.parts:
            p.set_async_io(True)
        return True

    def loci_staging(self, rb, fb, nj, nr, with_mask, cigar_ops_per_job=16):
        """Page-locked, REUSED input and output buffers for a stream of concatenated job tables (align_loci): JobTable.concat writes
        straight into them, the slices' copies in and out are asynchronous, and nothing is allocated or registered per batch once the
        buffers have grown to the largest batch seen (+25 %).  Returns the eight input arrays cut to the sizes asked for.  Same
        lifetime rule for results as pin_host."""
        if self._staging is None:
            self._staging = LociStaging(getattr(self.parts[0], "_L", None))
        arrays = self._staging(rb, fb, nj, nr, with_mask)
        self.ensure_outputs(nj, cigar_ops_per_job)
        return arrays

    def ensure_outputs(self, nj, cigar_ops_per_job=16):
        """two page-locked (records, CIGAR pool) pairs for batches of up to nj jobs, used in turn by collect(); the contexts switch to
        asynchronous transfers.  Nothing happens once they are large enough (they grow by a quarter beyond what is asked for)."""
        L = getattr(self.parts[0], "_L", None)
        if L is None or not hasattr(L, "ipx_pin_host"):
            return False
        if self._out is not None and self._out_loci and len(self._out[0][0]) >= nj:
            return True
        for a in self._out_pinned:
            L.ipx_unpin_host(C.c_void_p(a.ctypes.data))
        cap, k = int(nj * 1.25) + 1024, len(self.parts)
        out = [(np.empty(cap, RESULT_DTYPE), np.empty(cap * cigar_ops_per_job + 1024 * k, np.uint32)) for _ in (0, 1)]
        self._out_pinned = [a for a in out[0] + out[1] if L.ipx_pin_host(C.c_void_p(a.ctypes.data), a.nbytes) == 0]
        self._out, self._out_loci = out, True
        for p in self.parts:
            p.set_async_io(True)
        return True

    def unpin(self):
        if self._staging is not None:
            self._staging.close()
            self._staging = None
        L = getattr(self.parts[0], "_L", None) if self.parts else None
        for a in self._pinned + self._out_pinned:
            if L is not None:
                L.ipx_unpin_host(C.c_void_p(a.ctypes.data))
        self._pinned, self._out_pinned, self._out, self._out_loci = [], [], None, False
        for p in self.parts:
            if hasattr(p, "set_async_io") and getattr(p, "_ctx", None):
                p.set_async_io(False)

    def set_scoring(self, *a, **k):
        for p in self.parts:
            p.set_scoring(*a, **k)
        self.matrix = self.parts[0].matrix

    def set_routing(self, flags):
        for p in self.parts:
            p.set_routing(flags)

    def _worth_grouping(self, jobs):
        """a big batch of SEVERAL length classes (read length / 8, the 16-bit passes' class), each with a tenth of the jobs at least"""
        if self.group_by_length != "auto":
            return bool(self.group_by_length)
        if jobs.n_jobs < 400000 or len(self.parts) < 2:
            return False
        cls = np.bincount((np.diff(jobs.read_off) + 7) >> 3)
        return int((cls >= jobs.n_jobs // 10).sum()) >= 2

    def upload(self, jobs):
        """Make `jobs` resident: cut into one slice per stream and copied in.  A batch of mixed read lengths that is uploaded to be run (a batch
        that STAYS: run() many times, or upload / run / download of a large table) is first grouped by read length (JobTable.grouped_by_length;
        group_by_length = "auto" / True / False) and cut by WORK: a slice then holds one or two length classes, its launches are four times as
        large and half as many, and the classes are large enough for the banded reverse pass -- config 4: 58 -> 65 M aln/s.  download() returns
        the records in the caller's job order either way.  (submit() / align() -- a batch that passes through once -- do not: grouping a million
        jobs costs the host more than it saves the GPU.)"""
        self._order = None
        by_cells = self.balance_by_cells
        if self._worth_grouping(jobs):
            jobs, self._order = jobs.grouped_by_length()
            by_cells = True
        k = max(1, min(len(self.parts), jobs.n_jobs // self.min_jobs_per_stream))   # small batches: one stream
        b = shard_bounds(jobs.n_jobs, k, jobs if by_cells else None)
        self._active = self.parts[:k]
        self._slices = [jobs.shard(b[i], b[i + 1]) for i in range(k)]
        for p, j in zip(self._active, self._slices):
            p.upload(j)

    def run(self):
        for p in self._active:
            p.run()

    def sync(self):
        for p in self._active:
            p.sync()

    def download(self, cigar_ops_per_job=16):
        """Gather the slices' results.  Every slice downloads straight into its range of ONE record array and its own
        region of ONE cigar pool (offsets rebased in place), so nothing is concatenated; the pool may have unused gaps
        between the regions."""
        if self._order is not None:                                 # the batch was grouped by read length on its way in: back to the caller's order
            order, self._order = self._order, None
            try:
                got = self.download(cigar_ops_per_job)
            finally:
                self._order = order
            rec = np.empty_like(got.records)
            rec[order] = got.records
            return BatchResult(rec, got.cigar_pool)
        if not hasattr(self._active[0], "download_into"):          # test back-ends
            return merge_results([p.download(cigar_ops_per_job) for p in self._active])
        ns = [p._n_jobs for p in self._active]
        caps = [max(1024, n * cigar_ops_per_job) for n in ns]
        rec = np.empty(sum(ns), RESULT_DTYPE)
        pool = np.empty(sum(caps), np.uint32)
        lo = pb = 0
        for p, n, cap in zip(self._active, ns, caps):
            used = p.download_into(rec[lo:lo + n], pool[pb:pb + cap])
            if used < 0:                                           # a slice has more CIGAR ops than guessed: generic path
                return merge_results([q.download(cigar_ops_per_job) for q in self._active])
            rec["cigar_off"][lo:lo + n] += np.uint32(pb)
            lo += n
            pb += cap
        return BatchResult(rec, pool)

    def align(self, jobs):
        """upload + run + sync + download"""
        self.submit(jobs)
        return self.collect()

    def submit(self, jobs):
        """Cut `jobs` into slices and enqueue, per slice, the copies in and the whole pipeline; returns without waiting.
        collect() waits and fetches the results.  Two aligners used alternately (submit on one while collecting from the
        other) keep the GPU busy across batches: bench_modes.run_end_to_end."""
        k = max(1, min(len(self.parts), jobs.n_jobs // self.min_jobs_per_stream))   # small batches: one stream
        b = shard_bounds(jobs.n_jobs, k, jobs if self.balance_by_cells else None)
        self._active = self.parts[:k]
        self._submitted = jobs
        self._order = None

        def one(i):                                                # cut, enqueue the copies of and launch slice i
            j = jobs.shard(b[i], b[i + 1])
            self._active[i].upload(j)
            self._active[i].run()
            return j
        # (r04: one host thread per slice was tried -- four uploads at once, then four pipelines: 38.7 M aln/s for the many-loci stream against
        #  45.7 in turn, because in turn slice i computes while slice i+1 is still being cut and uploaded)
        self._slices = [one(i) for i in range(k)]

    def collect(self):
        jobs = self._submitted
        if self._out is not None and len(self._out[0][0]) >= jobs.n_jobs and hasattr(self._active[0], "download_async_into"):
            got = self._align_tail_async()
            if got is not None:
                return got
        try:
            self.sync()
        except IpxError:
            return merge_results([p.align(j) for p, j in zip(self._active, self._slices)])   # e.g. a slice outgrew its device cigar pool
        return self.download()

    def _align_tail_async(self, cigar_ops_per_job=16):
        """sync slice k, start its download into the pinned output buffers, go on to slice k+1: a slice's records travel while
        the later slices still compute.  None when something did not fit (the caller falls back to the blocking path)."""
        rec, pool = self._out[self._out_turn]
        rec = rec[:self._submitted.n_jobs]                       # (the buffers may be larger than this batch: loci_staging)
        self._out_turn ^= 1
        lo = pb = 0
        try:
            for p in self._active:
                p.sync()
                n = p._n_jobs
                cap = max(1024, n * cigar_ops_per_job)
                if pb + cap > len(pool) or p.download_async_into(rec[lo:lo + n], pool[pb:pb + cap]) < 0:
                    for q in self._active:
                        q.wait()
                    return None
                p._dl = (lo, n, pb)
                lo += n
                pb += cap
            for p in self._active:
                p.wait()
        except IpxError:
            for q in self._active:
                try:
                    q.wait()
                except IpxError:
                    pass
            return None
        for p in self._active:
            lo, n, pb = p._dl
            rec["cigar_off"][lo:lo + n] += np.uint32(pb)
        return BatchResult(rec, pool)

    def set_profiling(self, on):
        for p in self.parts:
            p.set_profiling(on)

    def kernel_times(self):
        out = {}
        for p in self._active:
            for name, (t, c) in p.kernel_times().items():
                t0, c0 = out.get(name, (0.0, 0))
                out[name] = (t0 + t, c0 + c)
        return out

    def kernel_units(self):
        out = {}
        for p in self._active:
            for name, u in p.kernel_units().items():
                out[name] = out.get(name, 0) + u
        return out

    def last_run_ms(self):
        return max(p.last_run_ms() for p in self._active)


def device_count():
    return int(_lib.lib().ipx_device_count())


def shard_bounds(n_jobs, n_shards, jobs=None):
    """Contiguous job ranges, shard k owns [b[k], b[k+1]).  Without a table: near-equal job COUNTS.  With the job table: near-equal
    WORK -- cumulative (read length + 40) x (window length + 60): the cell count of the forward pass plus what a job costs whatever its size
    (proofs, traceback, the wavefront's lead-in: config 4 grouped by length, 64.9 -> 66.1 M aln/s against plain cells) -- so that a table
    sorted by read length (the length-bucketed table of SURVEY.md 8e) does not hand the last shard three times the first one's cells."""
    if jobs is None or n_shards <= 1 or n_jobs == 0:
        return [n_jobs * k // n_shards for k in range(n_shards + 1)]
    cache = jobs.__dict__.setdefault("_work_bounds", {})           # (a table's cuts are asked for again every step it is aligned)
    if n_shards in cache:
        return list(cache[n_shards])
    cells = np.cumsum((np.diff(jobs.read_off).astype(np.float64) + 40.0) * (np.diff(jobs.ref_off)[jobs.ref_id] + 60.0))
    cuts = np.searchsorted(cells, cells[-1] * np.arange(1, n_shards) / n_shards, side="left") + 1
    b = [0] + [int(min(max(c, 0), n_jobs)) for c in cuts] + [n_jobs]
    for k in range(1, len(b)):                                     # monotone (degenerate tables: empty shards are fine)
        b[k] = max(b[k], b[k - 1])
    cache[n_shards] = tuple(b)
    return b


def align_sharded(jobs, aligners):
    """Split a job table over several GPUs (one host thread per GPU, no collective: every job is
    independent) and gather the records back in job order."""
    n = jobs.n_jobs
    k = len(aligners)
    b = shard_bounds(n, k)
    parts = [None] * k
    errs = []

    def work(i):
        try:
            parts[i] = aligners[i].align(jobs.shard(b[i], b[i + 1]))
        except Exception as e:      # surfaced to the caller below
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return merge_results(parts)


def merge_results(parts):
    """Host-side gather: concatenate shard records, rebasing cigar offsets into one pool.  (One part: handed back as it is.)"""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    if not parts:
        return BatchResult(np.zeros(0, RESULT_DTYPE), np.zeros(0, np.uint32))
    rec = np.concatenate([p.records for p in parts])
    lo = base = 0
    for p in parts:
        n = len(p.records)
        if base:
            rec["cigar_off"][lo:lo + n] += np.uint32(base)
        lo += n
        base += len(p.cigar_pool)
    return BatchResult(rec, np.concatenate([p.cigar_pool for p in parts]))


def align_loci(tables, match_score=2, mismatch_penalty=2, device=0, aligner=None):
    """MANY loci, one GPU batch: tables = one JobTable per locus (e.g. retarget_jobs / realign_pileup_jobs of each).  Returns one
    BatchResult per locus.  aligner: a MultiStreamAligner / GpuAligner to reuse (its scoring is left as it is); else the shared
    one of `device` with (match_score, mismatch_penalty)."""
    tables = list(tables)
    if aligner is None:
        from .sswpy import _gpu
        aligner = _gpu(device)
        aligner.set_scoring(matrix=dna_score_matrix(match_score, mismatch_penalty), flag=1, score_size=2)
    table = JobTable.concat(tables, staging=getattr(aligner, "loci_staging", None))
    return aligner.align(table).split(table.table_jobs if len(tables) else [])


def align_loci_stream(lists, match_score=2, mismatch_penalty=2, device=0, depth=2, streams=4, aligner_cls=None):
    """A STREAM of locus lists (a caller working through a VCF: one list of per-locus JobTables per chunk of rows): a generator that yields,
    for every list of `lists` and in order, what align_loci returns for it.  Three stages overlap: a host thread concatenates list k+1 into
    one of depth + 1 page-locked staging sets (JobTable.concat -> ipx_concat_tables: the library's copy, no GIL); the caller's thread
    enqueues list k on aligner k % depth (copies in, pipeline) and then collects list k-1 (copies out, split), so the GPU always has the
    next batch queued.  Two aligners of four streams are the eight hardware queues the loader asks the HIP runtime for (_lib.load); a
    third aligner -- twelve streams -- runs into queue sharing again (r04: 21.7 ms per 1.2 M-job list against 16.5).
    LIFETIME: the BatchResults yielded are views of the aligners' page-locked output buffers; they stay valid until 2 * depth - 1 more
    lists have been yielded (BatchResult.copy() detaches them)."""
    import queue
    import threading
    depth = max(1, depth)
    ring = [MultiStreamAligner(device, match_score, mismatch_penalty, streams=streams, aligner_cls=aligner_cls) for _ in range(depth)]
    stages = [LociStaging(getattr(ring[0].parts[0], "_L", None)) for _ in range(depth + 1)]
    free = [threading.Semaphore(1) for _ in stages]                # a staging set may be overwritten: the list made in it is collected
    ready = queue.Queue(maxsize=len(stages))
    stop = threading.Event()

    def feeder():
        try:
            for k, tables in enumerate(lists):
                s = k % len(stages)
                while not free[s].acquire(timeout=0.1):
                    if stop.is_set():
                        return
                tables = tables if isinstance(tables, (list, tuple)) else list(tables)
                ready.put((k, JobTable.concat(tables, staging=stages[s]) if tables else None))
            ready.put((None, None))
        except BaseException as e:                                 # (the consumer re-raises it)
            ready.put((None, e))

    th = threading.Thread(target=feeder, daemon=True)
    th.start()

    def finish(pending):
        k, table = pending
        out = ring[k % depth].collect().split(table.table_jobs) if table is not None else []
        free[k % len(stages)].release()
        return out
    try:
        pending = None                                             # (list number, table) submitted, not collected
        while True:
            k, table = ready.get()
            if k is None:
                if isinstance(table, BaseException):
                    raise table
                break
            if table is not None:
                a = ring[k % depth]
                a.ensure_outputs(table.n_jobs)
                a.submit(table)
            if pending is not None:
                yield finish(pending)
            pending = (k, table)
        if pending is not None:
            yield finish(pending)
    finally:
        stop.set()
        th.join(timeout=5)
        for r in ring:
            r.close()
        for st in stages:
            st.close()
