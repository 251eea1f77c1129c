"""Host-side mirror of the reference interface (SSW / make_aligner / align / align_pileup), job
tables, sharding and gather -- on CPU, with the emulator standing in for the GPU context."""
import numpy as np
import pytest

import indelpost_amd as ip
from indelpost_amd import sswpy, localn
from indelpost_amd.batch import JobTable, merge_results, shard_bounds


@pytest.fixture()
def emu_as_gpu(emu, monkeypatch):
    cache = {}

    def fake(device=0):
        if device not in cache:
            cache[device] = emu(device)
        return cache[device]
    monkeypatch.setattr(sswpy, "_gpu", fake)
    monkeypatch.setattr(localn, "_gpu", fake)
    return fake


def test_ssw_class_matches_reference_binding(emu_as_gpu, golden_sswpy):
    """Alignment tuples captured from the reference's sswpy.SSW (first 40 vectors: all KATs of
    SURVEY 8c incl. lower case, U->A, start_idx/end_idx, gap_open=len)."""
    for k, c in enumerate(golden_sswpy[:40]):
        a = ip.SSW(c["match"], c["mismatch"])
        a.setReference(c["ref"])
        a.setRead(c["read"])
        assert list(a.align(**c["kwargs"])) == c["expect"], "case %d" % k


def test_ssw_bytes_and_errors(emu_as_gpu):
    a = ip.SSW(3, 2)
    with pytest.raises(ValueError):
        a.align()                                   # call setReference first (sswpy.pyx:279-280)
    a.setReference(b"ACGTACGTTTGACCAGT")
    a.setRead(b"ACGTAGTTTGACCAGT")
    assert tuple(a.align(3, 1)) == ("5M1D11M", 45, 3, 0, 16, 0, 15)
    with pytest.raises(ValueError):
        a.align(start_idx=-1)
    with pytest.raises(ValueError):
        a.align(end_idx=100)
    assert ip.dna_score_matrix(2, 2).reshape(5, 5).tolist() == [[2, -2, -2, -2, 0], [-2, 2, -2, -2, 0],
                                                               [-2, -2, 2, -2, 0], [-2, -2, -2, 2, 0], [0] * 5]


def test_make_aligner_align_and_batch(emu_as_gpu):
    ref = "ACGTTGCATGCCGATAGGCTTAACGGATCGATCGGGATTACAGCTAGCTAG"
    reads = ["GCATGCCGATAGGCTTAACGG", "GATCGATCGGATTACAGC", "TTTTTTTT", "NNNNN", "gcatgccgataggcttaacgg"]
    al = ip.make_aligner(ref, 3, 2)
    one_by_one = [ip.align(al, r, 3, 1) for r in reads]
    assert al.align_batch(reads, 3, 1) == one_by_one
    assert one_by_one[0] == one_by_one[4]
    pairs = ip.align_pileup(reads, ref[:20] + ref[26:], ref, 3, 2, 3, 1)
    ref_al, mut_al = ip.make_aligner(ref, 3, 2), ip.make_aligner(ref[:20] + ref[26:], 3, 2)
    for r, (ra, ma) in zip(reads, pairs):
        assert ra == ip.align(ref_al, r, 3, 1)
        assert ma == ip.align(mut_al, r, len(r), 1)           # localn.pyx:253-255


def test_vectorised_host_paths_equal_the_per_item_ones(emu):
    """cigar_strings() / alignments_from() / realign_pileup_jobs() do in bulk what the per-alignment code does"""
    from indelpost_amd import localn, sswpy
    from indelpost_amd.batch import encode_dna
    ref = "ACGTTGCATGCCGATAGGCTTAACGGATCGATCGGGATTACAGCTAGCTAG"
    mut = ref[:20] + ref[26:]
    reads = ["GCATGCCGATAGGCTTAACGG", b"GATCGATCGGATTACAGC", "TTTTTTTT", "NNNNN", "A", "gcatgccgataggcttaacgg" * 3]
    jobs = localn.realign_pileup_jobs(reads, mut, ref, 3, 1)
    enc = [encode_dna(r) for r in reads]
    slow = JobTable.from_sequences([e for e in enc for _ in (0, 1)], [encode_dna(ref), encode_dna(mut)], [0, 1] * len(reads),
                                   [v for e in enc for v in (3, len(e))], 1, encoded=True)
    for f in ("reads", "read_off", "refs", "ref_off", "ref_id", "gap_open", "gap_ext"):
        assert np.array_equal(getattr(jobs, f), getattr(slow, f)), f
    res = emu(0, 3, 2).align(jobs)
    assert res.cigar_strings() == [res.cigar_string(i) for i in range(len(res))]
    assert sswpy.alignments_from(res) == [sswpy._alignment_from(res, i) for i in range(len(res))]
    assert localn.realign_pileup_jobs([], mut, ref, 3, 1).n_jobs == 0


def test_gap_penalties_narrow_to_uint8():
    j = JobTable.from_sequences(["ACGT"], ["ACGT"], [0], 300, 256 + 7)
    assert j.gap_open[0] == 300 - 256 and j.gap_ext[0] == 7  # ssw.h:129-130


def test_shard_and_merge(emu):
    rng = np.random.default_rng(1)
    refs = [rng.integers(0, 4, 120).astype(np.int8) for _ in range(3)]
    reads = [np.resize(refs[i % 3][i:], 40 + i).copy() for i in range(11)]
    jobs = JobTable.from_sequences(reads, refs, [i % 3 for i in range(11)], 3, 1, encoded=True)
    whole = emu(0, 3, 2).align(jobs)
    for k in (1, 2, 3, 4):
        b = shard_bounds(jobs.n_jobs, k)
        assert b[0] == 0 and b[-1] == 11 and all(b[i] <= b[i + 1] for i in range(k))
        parts = [emu(i, 3, 2).align(jobs.shard(b[i], b[i + 1])) for i in range(k)]
        merged = merge_results(parts)
        for i in range(11):
            assert merged.as_dict(i) == whole.as_dict(i)


def test_align_sharded_threads(emu):
    rng = np.random.default_rng(2)
    w = rng.integers(0, 4, 200).astype(np.int8)
    reads = [w[i:i + 60].copy() for i in range(0, 90, 10)]
    jobs = JobTable.from_sequences(reads, [w], [0] * len(reads), 3, 1, encoded=True)
    whole = emu(0, 3, 2).align(jobs)
    merged = ip.align_sharded(jobs, [emu(0, 3, 2), emu(1, 3, 2)])
    assert all(merged.as_dict(i) == whole.as_dict(i) for i in range(len(reads)))


def test_multi_stream_aligner_matches_single(emu):
    rng = np.random.default_rng(4)
    refs = [rng.integers(0, 4, 150).astype(np.int8) for _ in range(2)]
    reads = [np.resize(refs[i % 2][i:], 40 + (i % 30)).copy() for i in range(200)]
    jobs = JobTable.from_sequences(reads, refs, [i % 2 for i in range(200)], 3, 1, encoded=True)
    whole = emu(0, 3, 2).align(jobs)
    ms = ip.MultiStreamAligner(0, 3, 2, streams=3, aligner_cls=emu)
    ms.min_jobs_per_stream = 50
    got = ms.align(jobs)
    assert all(got.as_dict(i) == whole.as_dict(i) for i in range(200))
    small = JobTable.from_sequences(reads[:5], refs, [0] * 5, 3, 1, encoded=True)
    assert len(ms.align(small)) == 5          # too small to split: one stream


def test_out_of_scope_shells_say_so():
    """the API names the reference exports (indelpost/__init__.py:1-8) exist with its constructor signatures"""
    import inspect
    assert list(inspect.signature(ip.Variant.__init__).parameters)[1:] == ["chrom", "pos", "ref", "alt", "reference", "skip_validation"]
    va = inspect.signature(ip.VariantAlignment.__init__).parameters
    assert list(va)[1:4] == ["target", "bam", "window"] and va["window"].default == 50 and va["match_score"].default == 3 and \
        va["gap_open_penalty"].default == 3 and va["gap_extension_penalty"].default == 1 and va["downsample_threshold"].default == 1000
    with pytest.raises(NotImplementedError):
        ip.VariantAlignment(None, None)
    with pytest.raises(NotImplementedError):
        ip.Contig(None, [], None, 20, 1)
    with pytest.raises(NotImplementedError):
        ip.FailedContig()

    class Fa:
        def fetch(self, c, a, b):
            return "ACGT"[a % 4:a % 4 + 1]
    nv = ip.NullVariant("1", 7, Fa())
    assert not nv and nv.ref == nv.alt == "G" and nv.pos == 7


def test_shard_paths_keep_every_job_intact():
    """JobTable.shard: the three window-shipping paths (all windows, dense id range, gathered) give tables whose
    jobs see the same read, window and penalties as in the parent table."""
    rng = np.random.default_rng(6)
    n_refs, n_jobs = 600, 900
    refs = [rng.integers(0, 4, int(rng.integers(1500, 2600))).astype(np.int8) for _ in range(n_refs)]   # > 1 MiB of windows
    reads = [rng.integers(0, 4, int(rng.integers(5, 40))).astype(np.int8) for _ in range(n_jobs)]

    def check(jobs, lo, hi):
        sh = jobs.shard(lo, hi)
        assert sh.n_jobs == hi - lo
        for k in range(0, hi - lo, 7):
            j = lo + k
            assert np.array_equal(sh.reads[sh.read_off[k]:sh.read_off[k + 1]], jobs.reads[jobs.read_off[j]:jobs.read_off[j + 1]])
            a, b = sh.ref_id[k], jobs.ref_id[j]
            assert np.array_equal(sh.refs[sh.ref_off[a]:sh.ref_off[a + 1]], jobs.refs[jobs.ref_off[b]:jobs.ref_off[b + 1]])
            assert sh.gap_open[k] == jobs.gap_open[j] and sh.gap_ext[k] == jobs.gap_ext[j]
        return sh

    go = rng.integers(0, 9, n_jobs)
    sorted_ids = np.sort(rng.integers(0, n_refs, n_jobs))                     # a table sorted by locus: dense id ranges
    jobs = JobTable.from_sequences(reads, refs, sorted_ids, go, 1, encoded=True)
    sh = check(jobs, 100, 400)
    assert sh.n_refs < n_refs and len(sh.refs) < len(jobs.refs)                # only the range travelled
    scattered = rng.integers(0, n_refs, n_jobs)
    scattered[::2] = np.where(scattered[::2] % 2 == 0, 0, n_refs - 1)          # ids span everything, most windows unused
    jobs2 = JobTable.from_sequences(reads, refs, scattered, go, 1, encoded=True)
    sh2 = check(jobs2, 0, 60)
    assert sh2.n_refs <= 60 and len(sh2.refs) < len(jobs2.refs) // 4           # gathered
    assert check(jobs2, 10, 10).n_jobs == 0
    small = JobTable.from_sequences(reads[:20], refs[:3], [i % 3 for i in range(20)], 3, 1, encoded=True)
    assert check(small, 5, 15).n_refs == 3                                     # small window set: shipped whole


def test_result_digest_agrees_with_the_cpu_checker(emu, oracle_mod):
    """BatchResult.digest() (what bench.py compares with tests/golden/bench_digests.json) equals the digest computed
    from the CPU checker's records on the same table, and notices a single changed CIGAR op."""
    from indelpost_amd.batch import record_digest
    rng = np.random.default_rng(9)
    w = rng.integers(0, 4, 220).astype(np.int8)
    reads = []
    for i in range(40):
        r = w[i * 3:i * 3 + 70].copy()
        r[rng.integers(0, 70)] ^= 1
        if i % 4 == 0:
            r = np.concatenate([r[:30], r[34:]])
        reads.append(r)
    jobs = JobTable.from_sequences(reads, [w], [0] * 40, 3, 1, encoded=True)
    res = emu(0, 3, 2).align(jobs)
    rec = oracle_mod.cpu_batch_results(oracle_mod.Backend("port"), jobs, oracle_mod.dna_matrix(3, 2), 2)
    assert res.digest() == record_digest(rec, rec["cigar_hash"])
    i = int(np.flatnonzero(res.records["cigar_len"] > 1)[0])
    res.cigar_pool[int(res.records["cigar_off"][i]) + 1] += 16
    assert res.digest() != record_digest(rec, rec["cigar_hash"])


def test_synthetic_config_tables(hip_lib):
    """the multi-window generator behind bench.py's config-4 / config-5 shapes: deterministic, shapes as SURVEY 8d says"""
    from indelpost_amd import synth
    a, b = synth.config4_jobs(n_windows=5, reads_per_window=60), synth.config4_jobs(n_windows=5, reads_per_window=60)
    assert a.n_jobs == 300 and a.n_refs == 5 and np.array_equal(a.reads, b.reads) and np.array_equal(a.refs, b.refs)
    wl = np.diff(a.ref_off)
    assert wl.min() >= 200 and wl.max() <= 600
    lens = np.diff(a.read_off).reshape(5, 6, 10)
    for k, rl in enumerate((75, 100, 125, 150, 200, 250)):
        assert (lens[:, k, :] == np.minimum(rl, wl)[:, None]).all()
    assert (a.ref_id == np.repeat(np.arange(5), 60)).all() and a.reads.min() >= 0 and a.reads.max() <= 3
    c = synth.config5_jobs(n_loci=3, reads_per_locus=4)
    assert c.n_jobs == 72 and c.n_refs == 12 and (np.diff(c.ref_off) == 300).all() and (np.diff(c.read_off) == 150).all()
    assert [tuple(x) for x in np.stack([c.gap_open[:6], c.gap_ext[:6]], 1).tolist()] == synth.PENALTY_GRID
    assert (c.ref_id == np.repeat(np.arange(12), 6)).all()
    assert np.array_equal(c.reads[:150], c.reads[150 * 5:150 * 6]) and not np.array_equal(c.reads[:150], c.reads[900:1050])


def test_shard_bounds_by_work_balances_a_length_sorted_table():
    """shard_bounds with the table cuts by cumulative work -- (read length + 40) x (window length + 60): the forward pass's cells plus what a
    job costs whatever its size -- : on a table sorted by read length (SURVEY.md 8e's length-bucketed table) equal job COUNTS would give
    the last of 8 shards several times the first one's work"""
    rng = np.random.default_rng(3)
    lens = np.sort(rng.choice([75, 100, 125, 150, 200, 250], 4000))
    wins = [rng.integers(0, 4, int(n)).astype(np.int8) for n in rng.integers(200, 601, 40)]
    rid = rng.integers(0, 40, 4000).astype(np.int32)
    jobs = JobTable.from_sequences([np.zeros(int(n), np.int8) for n in lens], wins, rid, 3, 1, encoded=True)
    cells = (np.diff(jobs.read_off) + 40) * (np.diff(jobs.ref_off)[jobs.ref_id] + 60)
    for k in (2, 4, 8):
        b = shard_bounds(jobs.n_jobs, k, jobs)
        assert b[0] == 0 and b[-1] == jobs.n_jobs and all(x <= y for x, y in zip(b, b[1:]))
        w = np.array([cells[b[i]:b[i + 1]].sum() for i in range(k)], np.float64)
        assert w.max() / w.mean() < 1.05 and w.min() / w.mean() > 0.95, w
        byc = shard_bounds(jobs.n_jobs, k)
        wc = np.array([cells[byc[i]:byc[i + 1]].sum() for i in range(k)], np.float64)
        assert wc.max() / wc.min() > 1.5                              # what equal counts would have done
    assert shard_bounds(0, 4, jobs.shard(0, 0)) == [0, 0, 0, 0, 0]


def test_concat_and_split_round_trip(emu):
    """the many-loci entry: JobTable.concat of per-locus tables = the jobs in order with windows renumbered; BatchResult.split hands
    every locus its own records back, equal to aligning the locus alone"""
    from indelpost_amd.batch import align_loci
    rng = np.random.default_rng(12)
    tables = []
    for k in range(6):
        wins = [rng.integers(0, 4, int(rng.integers(60, 120))).astype(np.int8) for _ in range(int(rng.integers(1, 3)))]
        reads, rid = [], []
        for i in range(int(rng.integers(0, 5))):
            w = int(rng.integers(0, len(wins)))
            st = int(rng.integers(0, 20))
            r = wins[w][st:st + 40].copy()
            r[int(rng.integers(0, 40))] ^= 2
            reads.append(r); rid.append(w)
        tables.append(JobTable.from_sequences(reads, wins, np.array(rid, np.int32), 3, 1, encoded=True))
    a = emu(0, 3, 2)
    parts = align_loci(tables, aligner=a)
    assert [len(p) for p in parts] == [t.n_jobs for t in tables]
    for t, p in zip(tables, parts):
        if t.n_jobs:
            alone = a.align(t)
            assert all(p.as_dict(i) == alone.as_dict(i) for i in range(t.n_jobs))


def test_stream_of_locus_lists(emu):
    """align_loci_stream: a generator over lists of per-locus tables -- concatenation on a host thread, submit / collect on the caller's,
    a ring of three aligners -- yields, list by list and in order, what align_loci gives for each; empty lists and empty tables included;
    an exception in the producer surfaces in the consumer; abandoning the generator closes the aligners"""
    from indelpost_amd.batch import align_loci, align_loci_stream
    rng = np.random.default_rng(77)

    def locus():
        wins = [rng.integers(0, 4, int(rng.integers(60, 120))).astype(np.int8) for _ in range(int(rng.integers(1, 3)))]
        reads, rid = [], []
        for i in range(int(rng.integers(0, 5))):
            w = int(rng.integers(0, len(wins)))
            st = int(rng.integers(0, 20))
            r = wins[w][st:st + 40].copy()
            r[int(rng.integers(0, 40))] ^= 2
            reads.append(r); rid.append(w)
        return JobTable.from_sequences(reads, wins, np.array(rid, np.int32), 3, 1, encoded=True)
    lists = [[locus() for _ in range(int(rng.integers(1, 6)))] for _ in range(7)]
    lists.insert(3, [])
    one = emu(0, 3, 2)
    n = 0
    for tables, parts in zip(lists, align_loci_stream(iter(lists), 3, 2, streams=2, aligner_cls=emu)):
        want = align_loci(tables, aligner=one) if tables else []
        assert [len(p) for p in parts] == [t.n_jobs for t in tables]
        for p, q in zip(parts, want):
            assert all(p.as_dict(i) == q.as_dict(i) for i in range(len(q)))
        n += 1
    assert n == len(lists)

    def broken():
        yield lists[0]
        raise ValueError("no more rows")
    gen = align_loci_stream(broken(), 3, 2, streams=2, aligner_cls=emu)
    import pytest
    with pytest.raises(ValueError):
        for _ in gen:
            pass
    gen = align_loci_stream(iter(lists), 3, 2, streams=2, aligner_cls=emu)
    next(gen)
    gen.close()                                                      # (the feeder thread stops, the ring is closed)


def test_pinned_output_bookkeeping_of_the_multi_stream_aligner(emu):
    """collect() with page-locked output buffers (here: plain arrays handed in as if pin_host had made them): every slice is copied
    into its range of ONE record array and its region of ONE cigar pool, cigar offsets rebased; two output pairs are used in turn, so
    the result of one batch survives the collection of the next; a pool too small for a slice falls back to the blocking path"""
    from indelpost_amd._lib import RESULT_DTYPE
    rng = np.random.default_rng(31)
    w = rng.integers(0, 4, 200).astype(np.int8)

    def table(seed):
        r2 = np.random.default_rng(seed)
        reads = []
        for i in range(10):
            st = int(r2.integers(0, 120))
            r = w[st:st + 60].copy()
            r[int(r2.integers(0, 60))] ^= 1
            if i % 3 == 0:
                r = np.concatenate([r[:25], r[28:]])
            reads.append(r)
        return JobTable.from_sequences(reads, [w], [0] * 10, 3, 1, encoded=True)
    a, b = table(1), table(2)
    m = ip.MultiStreamAligner(0, 3, 2, streams=2, aligner_cls=emu)
    m.min_jobs_per_stream = 4                                            # two slices of five jobs
    plain_a, plain_b = m.align(a), m.align(b)                            # blocking path (no pinned outputs)
    m._out = [(np.zeros(10, RESULT_DTYPE), np.zeros(4096, np.uint32)) for _ in (0, 1)]
    got_a = m.align(a)
    got_b = m.align(b)                                                   # the other output pair: got_a stays valid
    for got, plain in ((got_a, plain_a), (got_b, plain_b)):
        assert all(got.as_dict(i) == plain.as_dict(i) for i in range(10))
    assert np.shares_memory(got_a.records, m._out[0][0]) and np.shares_memory(got_b.records, m._out[1][0])     # (views: the buffers may be larger than a batch, loci_staging)
    assert got_a.records["cigar_off"][5] >= 1024                         # second slice: rebased into its region of the pool
    kept = got_a.copy()
    m.align(b)                                                           # third collect: overwrites the first pair ...
    assert all(kept.as_dict(i) == plain_a.as_dict(i) for i in range(10))   # ... the copy is unaffected
    m._out = [(np.zeros(10, RESULT_DTYPE), np.zeros(1500, np.uint32)) for _ in (0, 1)]   # room for one slice's region only
    small = m.align(a)                                                   # falls back to the blocking download
    m._out = [(np.zeros(25, RESULT_DTYPE), np.zeros(8192, np.uint32)) for _ in (0, 1)]   # buffers LARGER than the batch (reused staging): views of the first 10 records
    big = m.align(a)
    assert len(big.records) == 10 and np.shares_memory(big.records, m._out[0][0]) and all(big.as_dict(i) == plain_a.as_dict(i) for i in range(10))
    assert all(small.as_dict(i) == plain_a.as_dict(i) for i in range(10)) and small.records is not m._out[0][0]
    m.close()


def test_record_digest_is_xxh64(hip_lib):
    """ipx_record_digest (the library's host-side XXH64 over ten int64 per job) equals the `xxhash` package's XXH64 of the same bytes --
    the definition tests/golden/bench_digests.json was generated under -- for every stripe remainder (n jobs x 80 bytes mod 32)"""
    xxhash = pytest.importorskip("xxhash")
    from indelpost_amd.batch import record_digest
    from indelpost_amd._lib import RESULT_DTYPE
    rng = np.random.default_rng(31)
    for n in (0, 1, 2, 3, 4, 5, 7, 64, 1001):
        rec = np.zeros(n, RESULT_DTYPE)
        for f in ("score1", "score2", "cigar_len"):
            rec[f] = rng.integers(0, 65536, n)
        for f in ("ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2"):
            rec[f] = rng.integers(-1, 40000, n)
        rec["flag"] = rng.integers(0, 3, n)
        ch = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        m = np.empty((n, 10), np.int64)
        for k, f in enumerate(("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "flag", "cigar_len")):
            m[:, k] = rec[f]
        m[:, 9] = ch
        assert record_digest(rec, ch) == xxhash.xxh64(np.ascontiguousarray(m, "<i8").tobytes()).intdigest(), n


def test_library_concat_handles_views_masks_and_empty_tables(hip_lib):
    """ipx_concat_tables behind JobTable.concat: tables that are VIEWS into a bigger table (offsets not starting at 0), tables with and
    without mask_len, an empty table and a table without windows; equal to a plain numpy concatenation; a bad ref_id is refused"""
    rng = np.random.default_rng(77)
    refs = [rng.integers(0, 4, int(n)).astype(np.int8) for n in (40, 77, 5, 120)]
    reads = [rng.integers(0, 5, int(rng.integers(0, 60))).astype(np.int8) for _ in range(23)]
    big = JobTable.from_sequences(reads, refs, [i % 4 for i in range(23)], [3 + i % 5 for i in range(23)], [i % 2 for i in range(23)], encoded=True)
    parts = [big.shard(0, 7), big.shard(7, 7), big.shard(7, 19), big.shard(19, 23),
             JobTable(np.zeros(0, np.int8), np.zeros(1, np.int64), np.zeros(0, np.int8), np.zeros(1, np.int64), np.zeros(0, np.int32), 0, 0)]
    # a table whose arrays are longer than its offsets say, starting at an offset of its own
    t = big.shard(3, 9)
    t2 = JobTable(np.concatenate([np.full(5, 9, np.int8), t.reads, np.full(3, 9, np.int8)]), t.read_off + 5, t.refs, t.ref_off, t.ref_id, t.gap_open, t.gap_ext)
    parts.append(t2)
    cat = JobTable.concat(parts)
    assert cat.table_jobs.tolist() == [p.n_jobs for p in parts] and cat.mask_len is None
    at_job = at_ref = 0
    for p in parts:
        for k in range(p.n_jobs):
            lo, hi = int(p.read_off[k]), int(p.read_off[k + 1])
            clo, chi = int(cat.read_off[at_job + k]), int(cat.read_off[at_job + k + 1])
            assert np.array_equal(cat.reads[clo:chi], p.reads[lo:hi])
            w, cw = int(p.ref_id[k]), int(cat.ref_id[at_job + k])
            assert cw == w + at_ref
            assert np.array_equal(cat.refs[int(cat.ref_off[cw]):int(cat.ref_off[cw + 1])], p.refs[int(p.ref_off[w]):int(p.ref_off[w + 1])])
            assert cat.gap_open[at_job + k] == p.gap_open[k] and cat.gap_ext[at_job + k] == p.gap_ext[k]
        at_job += p.n_jobs
        at_ref += p.n_refs
    assert cat.n_jobs == at_job and cat.n_refs == at_ref and cat.read_off[-1] == len(cat.reads) and cat.ref_off[-1] == len(cat.refs)
    # mask_len survives only when every table has it
    m1 = JobTable(parts[0].reads, parts[0].read_off, parts[0].refs, parts[0].ref_off, parts[0].ref_id, parts[0].gap_open, parts[0].gap_ext, np.arange(7, dtype=np.int32) + 15)
    m2 = JobTable(parts[3].reads, parts[3].read_off, parts[3].refs, parts[3].ref_off, parts[3].ref_id, parts[3].gap_open, parts[3].gap_ext, np.arange(4, dtype=np.int32) + 40)
    both = JobTable.concat([m1, m2])
    assert both.mask_len.tolist() == list(range(15, 22)) + list(range(40, 44))
    assert JobTable.concat([m1, parts[3]]).mask_len is None
    assert JobTable.concat([]).n_jobs == 0
    bad = JobTable(parts[0].reads, parts[0].read_off, parts[0].refs, parts[0].ref_off, parts[0].ref_id + 50, parts[0].gap_open, parts[0].gap_ext)
    with pytest.raises(ip.IpxError):
        JobTable.concat([parts[3], bad])


def test_grouping_by_read_length(emu):
    """JobTable.grouped_by_length (the library's ipx_group_by_length): a stable counting sort of the jobs by read length -- reads, offsets and the
    per-job arrays in the new order, windows shared; and MultiStreamAligner.upload() with group_by_length on: the slices hold the grouped jobs,
    download() hands the records back in the caller's order, equal to the ungrouped run"""
    rng = np.random.default_rng(8)
    wins = [rng.integers(0, 4, int(n)).astype(np.int8) for n in (90, 140, 200)]
    reads, rid, go, ge, mask = [], [], [], [], []
    for i in range(60):
        ln = int(rng.choice([30, 30, 47, 64, 64, 64, 100, 0]))
        w = int(rng.integers(0, 3))
        st = int(rng.integers(0, max(1, len(wins[w]) - ln))) if ln else 0
        r = wins[w][st:st + ln].copy()
        if ln > 10:
            r[int(rng.integers(0, ln))] ^= 1
        reads.append(r); rid.append(w); go.append(int(rng.integers(2, 6))); ge.append(int(rng.integers(0, 2))); mask.append(int(rng.integers(15, 40)))
    jobs = JobTable.from_sequences(reads, wins, rid, go, ge, encoded=True)
    jobs.mask_len = np.asarray(mask, np.int32)
    g, order = jobs.grouped_by_length()
    lens = np.diff(jobs.read_off)
    assert order.tolist() == np.argsort(lens, kind="stable").tolist()
    assert np.diff(g.read_off).tolist() == sorted(lens.tolist()) and g.refs is jobs.refs
    for k, j in enumerate(order.tolist()):
        assert g.reads[g.read_off[k]:g.read_off[k + 1]].tobytes() == jobs.reads[jobs.read_off[j]:jobs.read_off[j + 1]].tobytes()
        assert (g.ref_id[k], g.gap_open[k], g.gap_ext[k], g.mask_len[k]) == (jobs.ref_id[j], jobs.gap_open[j], jobs.gap_ext[j], jobs.mask_len[j])
    jobs.mask_len = None
    plain = ip.MultiStreamAligner(0, 3, 2, streams=3, aligner_cls=emu)
    plain.min_jobs_per_stream = 10
    plain.group_by_length = False
    plain.upload(jobs); plain.run(); plain.sync()
    want = plain.download()
    m = ip.MultiStreamAligner(0, 3, 2, streams=3, aligner_cls=emu)
    m.min_jobs_per_stream = 10
    assert not m._worth_grouping(jobs)                                 # ("auto": a small batch is left as it is)
    m.group_by_length = True
    m.upload(jobs); m.run(); m.sync()
    assert m._order is not None and [int(np.diff(s_.read_off).max(initial=0)) for s_ in m._slices] == sorted(int(np.diff(s_.read_off).max(initial=0)) for s_ in m._slices)
    got = m.download()
    assert all(got.as_dict(i) == want.as_dict(i) for i in range(jobs.n_jobs))
    assert all(m.align(jobs).as_dict(i) == want.as_dict(i) for i in range(jobs.n_jobs)) and m._order is None      # (a batch that passes through: as given)
