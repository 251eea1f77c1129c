"""A pysam-free reader for the two file types the pileup front-end touches: BAM (BGZF) and FASTA.

indelPost takes a ``pysam.AlignmentFile`` and a ``pysam.FastaFile`` (SURVEY.md 8b lists what it touches:
``bam.fetch(chrom, start, end, until_eof=True)``, ``bam.count(..., read_callback=)``, ``bam.references``; per segment
``cigarstring, reference_start, reference_end, query_sequence, query_qualities, is_duplicate, is_secondary, is_reverse,
query_name, mapping_quality``; ``fasta.fetch / get_reference_length / references / filename``).  pysam is not available
where this package is built, so these classes provide exactly that surface in plain Python (zlib + struct): enough to
run the pileup front-end (indelpost_amd.pileup.make_pileup) on real files, and -- with ``write_bam`` -- to make the
synthetic loci of the tests and of BASELINE configs[2].  Regions are served by a linear scan (no .bai); fine for the
locus-sized files this path sees, not a general BAM library.
"""
import array
import struct
import zlib

_CIGAR_OPS = "MIDNSHP=X"
_SEQ_CODE = "=ACMGRSVTWYHKDBN"
_REF_CONSUMING = frozenset("MDN=X")


class FastaFile:
    """fetch(chrom, start, end) (0-based, half open), get_reference_length, references, filename.  Source: a path to a
    FASTA file (read into memory) or a {name: sequence} dict."""

    def __init__(self, source):
        self.filename = source if isinstance(source, str) else None
        self._seqs = {}
        if isinstance(source, dict):
            self._seqs = dict(source)
        else:
            name, parts = None, []
            with open(source) as f:
                for line in f:
                    if line.startswith(">"):
                        if name is not None:
                            self._seqs[name] = "".join(parts)
                        name, parts = line[1:].split()[0], []
                    else:
                        parts.append(line.strip())
            if name is not None:
                self._seqs[name] = "".join(parts)
        self.references = list(self._seqs)

    def fetch(self, chrom, start, end):
        if chrom not in self._seqs:
            raise KeyError("sequence '%s' not present" % chrom)
        return self._seqs[chrom][max(0, start):max(0, end)]

    def get_reference_length(self, chrom):
        return len(self._seqs[chrom])


class AlignedSegment:
    """the attributes of pysam.AlignedSegment that dictize_read and fetch_reads use (pileup.pyx:136-200)"""
    __slots__ = ("query_name", "flag", "reference_id", "reference_name", "reference_start", "mapping_quality", "cigartuples",
                 "query_sequence", "query_qualities")

    def __init__(self, query_name, flag, reference_name, reference_start, mapping_quality, cigarstring, query_sequence,
                 query_qualities=None, reference_id=0):
        self.query_name, self.flag, self.reference_name, self.reference_id = query_name, flag, reference_name, reference_id
        self.reference_start, self.mapping_quality, self.query_sequence = reference_start, mapping_quality, query_sequence
        self.cigartuples = _parse_cigar(cigarstring) if isinstance(cigarstring, str) else cigarstring
        self.query_qualities = None if query_qualities is None else array.array("B", query_qualities)

    @property
    def cigarstring(self):
        if not self.cigartuples:
            return None
        return "".join("%d%s" % (n, _CIGAR_OPS[op]) for op, n in self.cigartuples)

    @property
    def reference_end(self):
        if not self.cigartuples:
            return None
        return self.reference_start + sum(n for op, n in self.cigartuples if _CIGAR_OPS[op] in _REF_CONSUMING)

    is_reverse = property(lambda s: bool(s.flag & 0x10))
    is_secondary = property(lambda s: bool(s.flag & 0x100))
    is_duplicate = property(lambda s: bool(s.flag & 0x400))
    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    is_qcfail = property(lambda s: bool(s.flag & 0x200))


def _parse_cigar(s):
    out, n = [], 0
    for ch in s or "":
        if ch.isdigit():
            n = n * 10 + ord(ch) - 48
        else:
            out.append((_CIGAR_OPS.index(ch), n))
            n = 0
    return out


def _bgzf_blocks(raw):
    """decompressed payloads of the BGZF blocks of a file image"""
    p = 0
    while p < len(raw):
        if raw[p:p + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % p)
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        q, bsize = p + 12, None
        while q < p + 12 + xlen:
            si1, si2, slen = raw[q], raw[q + 1], struct.unpack_from("<H", raw, q + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", raw, q + 4)[0] + 1
            q += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        yield zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
        p += bsize


class AlignmentFile:
    """references, fetch(chrom, start, end, until_eof=True), count(chrom, start, end, read_callback=)"""

    def __init__(self, path):
        self.filename = path
        with open(path, "rb") as f:
            data = b"".join(_bgzf_blocks(f.read()))
        if data[:4] != b"BAM\x01":
            raise ValueError("%s is not a BAM file" % path)
        l_text = struct.unpack_from("<i", data, 4)[0]
        self.text = data[8:8 + l_text].decode(errors="replace")
        p = 8 + l_text
        n_ref = struct.unpack_from("<i", data, p)[0]
        p += 4
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name = struct.unpack_from("<i", data, p)[0]
            self.references.append(data[p + 4:p + 4 + l_name - 1].decode())
            self.lengths.append(struct.unpack_from("<i", data, p + 4 + l_name)[0])
            p += 8 + l_name
        self._segments = []
        while p + 4 <= len(data):
            block = struct.unpack_from("<i", data, p)[0]
            rec = data[p + 4:p + 4 + block]
            p += 4 + block
            ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
            q = 32
            name = rec[q:q + l_name - 1].decode()
            q += l_name
            cig = [(v & 15, v >> 4) for v in struct.unpack_from("<%dI" % n_cig, rec, q)]
            q += 4 * n_cig
            packed = rec[q:q + (l_seq + 1) // 2]
            q += (l_seq + 1) // 2
            seq = "".join(_SEQ_CODE[b >> 4] + _SEQ_CODE[b & 15] for b in packed)[:l_seq]
            qual = rec[q:q + l_seq]
            quals = None if (l_seq and qual[0] == 0xFF) else qual
            self._segments.append(AlignedSegment(name, flag, self.references[ref_id] if ref_id >= 0 else None, pos, mapq, cig, seq,
                                                 quals, ref_id))

    def fetch(self, chrom=None, start=None, end=None, until_eof=False):
        for s in self._segments:
            if chrom is not None:
                if s.reference_name != chrom:
                    continue
                e = s.reference_end if s.reference_end is not None else s.reference_start + 1
                if start is not None and e <= start:
                    continue
                if end is not None and s.reference_start >= end:
                    continue
            yield s

    def count(self, chrom, start, end, read_callback="nofilter"):
        """reads overlapping the region; "all" skips unmapped, secondary, QC-fail and duplicate reads (pysam's filter)"""
        n = 0
        for s in self.fetch(chrom, start, end):
            if read_callback == "all" and (s.flag & (0x4 | 0x100 | 0x200 | 0x400)):
                continue
            n += 1
        return n


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bam(path, references, segments):
    """A minimal BAM writer for synthetic loci: references = [(name, length)], segments = AlignedSegment objects (sorted by
    the caller).  One BGZF block per 60 KB of payload plus the EOF marker block."""
    out = [b"BAM\x01"]
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in references)
    out.append(struct.pack("<i", len(text)) + text.encode())
    out.append(struct.pack("<i", len(references)))
    names = [r[0] for r in references]
    for name, length in references:
        out.append(struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length))
    for s in segments:
        cig = s.cigartuples or []
        seq = s.query_sequence or ""
        codes = [_SEQ_CODE.index(c.upper()) if c.upper() in _SEQ_CODE else 15 for c in seq]
        if len(codes) % 2:
            codes.append(0)
        packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
        qual = bytes(s.query_qualities) if s.query_qualities is not None else b"\xff" * len(seq)
        end = s.reference_end if s.reference_end is not None else s.reference_start + 1
        ref_id = names.index(s.reference_name) if s.reference_name in names else -1
        body = struct.pack("<iiBBHHHiiii", ref_id, s.reference_start, len(s.query_name) + 1, s.mapping_quality,
                           _reg2bin(s.reference_start, end), len(cig), s.flag, len(seq), -1, -1, 0)
        body += s.query_name.encode() + b"\0" + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cig) + packed + qual
        out.append(struct.pack("<i", len(body)) + body)
    payload = b"".join(out)
    with open(path, "wb") as f:
        for p in range(0, max(len(payload), 1), 60000):
            chunk = payload[p:p + 60000]
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            cdata = comp.compress(chunk) + comp.flush()
            f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cdata) + 25))
            f.write(cdata + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # BGZF EOF marker
