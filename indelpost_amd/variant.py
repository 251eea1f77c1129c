"""``Variant`` / ``NullVariant``: the VCF-style variant objects the callers of the aligner pass around.

Host-side mirror of indelpost/variant.pyx (citations into /root/reference/indelpost/): construction and validation
(:92-160), type properties (:162-215), normalised equality and hash (:218-246), normalize (:276-324),
generate_equivalents (:327-371), the complex-indel helpers (:374-401), flanks and repeat counting (:483-560),
is_non_complex_indel (:563-578) and decompose_complex_variant (:581-632, its one alignment through this package's
aligner).  ``reference`` is a pysam.FastaFile duck type (fetch, get_reference_length, references, filename) --
indelpost_amd.bamio.FastaFile is one.  Not provided: query_vcf (:404-480; needs pysam's VCF reader).

This is string logic around the hot path, outside SURVEY.md section 8's scope table; it is here because the batched
callers (retarget's candidate matching, the pileup front-end) need a Variant to hand back and to compare with.
Parity: pinned by vectors produced by the reference's own class body executed as text (oracle/gen_variant_golden.py).
"""
from .cigar import findall_indels, to_minimal_repeat_unit

_BASES = set("ACTGNatcgn")


def repeat_counter(query_seq, flank_seq):
    """whole copies of query_seq at the start of flank_seq (utilities.pyx:169-184)"""
    q = len(query_seq)
    if len(flank_seq) < q:
        return 0
    n = 0
    for i in range(0, len(flank_seq), q):
        if flank_seq[i:i + q] != query_seq:
            break
        n += 1
    return n


class NullVariant:
    """variant.pyx:9-60: returned when the target is not found; falsy; ref == alt == the reference base at the locus"""

    def __init__(self, chrom, pos, reference):
        self.chrom = chrom
        self.pos = pos
        self.ref = reference.fetch(chrom, pos - 1, pos)
        self.alt = self.ref
        self.reference = reference

    def __bool__(self):
        return False

    def __eq__(self, other):
        if isinstance(other, Variant):
            return False
        return (self.chrom, self.pos, self.ref, self.alt) == (other.chrom, other.pos, other.ref, other.alt)

    def __hash__(self):
        return hash((self.chrom, self.pos, self.ref, self.alt))


class Variant:
    """variant.pyx:62-632.  Equality holds between objects that are identical in normalised form."""

    def __init__(self, chrom, pos, ref, alt, reference, skip_validation=False):
        self._chrom = chrom
        self.pos = pos
        self.ref = ref
        self.alt = alt
        self.reference = reference
        if skip_validation:
            self.chrom = chrom
        else:
            self.chrom = self._format_chrom_name(chrom, reference)
            self._validate()

    @staticmethod
    def _format_chrom_name(chrom, reference):                       # variant.pyx:119-138
        """the name in the FASTA's own convention: with or without "chr", mitochondrion as M or MT"""
        names = reference.references
        bare = chrom.replace("chr", "")
        mt_style = any(n in ("chrMT", "MT") for n in names)
        if bare in ("M", "MT"):
            bare = "MT" if mt_style else "M"
        return ("chr" + bare) if names[0].startswith("chr") else bare

    def _validate(self):                                            # variant.pyx:140-160
        if not self.ref or not self.alt:
            raise ValueError("Allele may not be empty")
        if self.ref == self.alt:
            raise ValueError("Not a variant: reference allele and alternate allele may not be identical")
        if not set(self.ref) <= _BASES or not set(self.alt) <= _BASES:
            self.ref = "".join(b if b in _BASES else "N" for b in self.ref)
            self.alt = "".join(b if b in _BASES else "N" for b in self.alt)
        try:
            ok = bool(self.reference.fetch(self.chrom, self.pos - 1, self.pos))
        except Exception:
            ok = False
        if not ok:
            raise ValueError("The locus is not defined in the reference")

    def __getstate__(self):
        return (self.chrom, self.pos, self.ref, self.alt, self.reference.filename)

    # -- type ------------------------------------------------------------------------------------------------------
    @property
    def variant_type(self):
        r, a = len(self.ref), len(self.alt)
        return "I" if r < a else "D" if r > a else "S" if a == 1 else "M"

    @property
    def is_del(self):
        return self.variant_type == "D"

    @property
    def is_ins(self):
        return self.variant_type == "I"

    @property
    def is_indel(self):
        return self.is_ins or self.is_del

    @property
    def indel_seq(self):
        if self.is_ins:
            return self.alt[len(self.ref):]
        if self.is_del:
            return self.ref[len(self.alt):]
        return ""

    # -- identity --------------------------------------------------------------------------------------------------
    def __eq__(self, other):
        if isinstance(other, NullVariant):
            return False
        i, j = self.normalize(), other.normalize()
        chrom_eq = (i.chrom.replace("chr", "") == j.chrom.replace("chr", "")) or (i._chrom.replace("chr", "") == j._chrom.replace("chr", ""))
        return chrom_eq and i.pos == j.pos and j.ref.upper() == i.ref.upper() and i.alt.upper() == j.alt.upper()

    def __hash__(self):
        i = self.normalize() if self.is_indel else self
        return hash((i._chrom, i.pos, i.ref, i.alt))

    @property
    def is_leftaligned(self):
        if self.ref[-1].upper() != self.alt[-1].upper():
            return True
        if "N" in self.ref.upper() or "N" in self.alt.upper():
            return True
        return None

    @property
    def is_normalized(self):
        if self.is_leftaligned:
            return not (len(self.ref) > 1 and len(self.alt) and self.ref[0].upper() == self.alt[0].upper())
        return False

    def normalize(self, inplace=False):
        """left-align (up to 300 bases) and trim to the minimal representation (variant.pyx:276-324)"""
        where, r, a = self.pos, self.ref, self.alt
        upstream = self.reference.fetch(self.chrom, max(0, where - 301), where - 1)
        # roll left while both alleles end in the same (non-N) base: drop it, prepend the base before the variant
        steps = 0
        while steps < len(upstream) and r[-1].upper() == a[-1].upper() != "N":
            prev = upstream[len(upstream) - 1 - steps]
            r, a = prev + r[:-1], prev + a[:-1]
            where -= 1
            steps += 1
        # trim the shared prefix down to one padding base
        while min(len(r), len(a)) > 1 and r[0].upper() == a[0].upper():
            r, a, where = r[1:], a[1:], where + 1
        if inplace:
            self.pos, self.ref, self.alt = where, r, a
            return None
        return Variant(self.chrom, where, r, a, self.reference, skip_validation=True)

    def generate_equivalents(self):
        """the normalised object and its right-shifted equivalents (variant.pyx:327-371)"""
        first = self.normalize()
        found = [first]
        if not first.is_indel:
            return found
        limit = 300
        downstream = first._right_of_event(limit)
        where, r, a, insertion = first.pos, first.ref, first.alt, first.is_ins
        cand = first
        for k in range(limit):
            if not (self == cand):                 # (the reference's loop condition: stops after the first shift that is no equivalent)
                break
            nxt = downstream[k]
            if insertion:                          # slide the inserted string one base to the right
                r, a = a[1], a[1:] + nxt
            else:                                  # slide the deleted string
                a, r = r[1], r[1:] + nxt
            where += 1
            cand = Variant(self.chrom, where, r, a, self.reference, skip_validation=True)
            if self == cand:
                found.append(cand)
        return found

    def _clone(self, pos=None, ref=None, alt=None):
        return Variant(self.chrom, self.pos if pos is None else pos, self.ref if ref is None else ref,
                       self.alt if alt is None else alt, self.reference, skip_validation=True)

    def _generate_equivalents_private(self):
        """(variant.pyx:374-384) a complex indel is pinned at the start and at the end of the deleted sequence"""
        return self.generate_equivalents() if self.is_non_complex_indel() else [self._clone(), self._clone(pos=self.pos + len(self.ref))]

    def _get_indel_seq(self, how=None):
        """(variant.pyx:386-393)"""
        if self.is_non_complex_indel():
            return self.indel_seq
        return {"I": self.alt[1:], "D": self.ref[1:]}.get(how)

    def _reduce_complex_indel(self, to=None):
        """(variant.pyx:395-401) the insertion or the deletion half of a complex indel; NullVariant for a non-complex one"""
        if self.is_non_complex_indel():
            return NullVariant(self.chrom, self.pos, self.reference)
        if to == "I":
            return self._clone(ref=self.alt[0])
        if to == "D":
            return self._clone(alt=self.ref[0])
        return None

    def query_vcf(self, *a, **k):
        raise NotImplementedError("Variant.query_vcf (indelpost/variant.pyx:404) needs pysam's VCF reader; not part of this package")

    # -- flanks and repeats ----------------------------------------------------------------------------------------
    def _right_of_event(self, window):
        ref_lim = self.reference.get_reference_length(self.chrom)
        if self.is_non_complex_indel() and self.variant_type == "I":
            return self.reference.fetch(self.chrom, self.pos, min(self.pos + window, ref_lim))
        event_len = len(self.indel_seq) if (self.is_non_complex_indel() and self.variant_type == "D") else len(self.ref) - 1
        return self.reference.fetch(self.chrom, self.pos + event_len, min(self.pos + event_len + window, ref_lim))

    def left_flank(self, window=50, normalize=False):
        i = Variant(self.chrom, self.pos, self.ref, self.alt, self.reference, skip_validation=True) if normalize else self
        pos = i.pos if i.is_non_complex_indel() else i.pos - 1
        return i.reference.fetch(i.chrom, max(0, pos - window), pos)

    def right_flank(self, window=50, normalize=False):
        i = Variant(self.chrom, self.pos, self.ref, self.alt, self.reference, skip_validation=True) if normalize else self
        return i._right_of_event(window)

    def count_repeats(self, by_repeat_unit=True):
        seq = self.indel_seq if self.is_non_complex_indel() else self.alt
        if by_repeat_unit:
            seq = to_minimal_repeat_unit(seq)
        return repeat_counter(seq, self.left_flank()[::-1]) + repeat_counter(seq, self.right_flank())

    def is_non_complex_indel(self):
        i = self.normalize()
        if len(i.ref) == len(i.alt) or i.ref[0] != i.alt[0]:
            return False
        return len(i.ref if i.is_ins else i.alt) <= 1

    def decompose_complex_variant(self, match_score=3, mismatch_penalty=2, gap_open_penalty=4, gap_extension_penalty=0):
        """the non-complex variants a complex one decomposes into under a Smith-Waterman alignment of the mutated against
        the reference sequence, +-100 bases (variant.pyx:581-632)"""
        if self.is_non_complex_indel():
            return [self]
        from .localn import align, make_aligner
        v = self.normalize()
        flank = 100
        left_end = v.pos - 1                                   # 0-based end of the left flank
        right_start = left_end + len(v.ref)
        fa, chrom = self.reference, v.chrom
        upstream = fa.fetch(chrom, left_end - flank, left_end)
        mutated = upstream + v.alt + fa.fetch(chrom, right_start, right_start + flank)
        original = fa.fetch(chrom, left_end - flank, right_start + flank)
        aln = align(make_aligner(original, match_score, mismatch_penalty), mutated, gap_open_penalty, gap_extension_penalty)
        indels, snvs = findall_indels(aln, left_end + 1 - flank + aln.reference_start, original, mutated, report_snvs=True)
        parts = []
        for d in indels:
            pad = d["lt_ref"][-1]
            ref, alt = (pad + d["del_seq"], pad) if d["indel_type"] == "D" else (pad, pad + d["indel_seq"])
            parts.append(self._clone(pos=d["pos"], ref=ref, alt=alt))
        parts.extend(self._clone(pos=x["pos"], ref=x["ref"], alt=x["alt"]) for x in snvs)
        return parts

    def __repr__(self):
        return "Variant(%r, %d, %r, %r)" % (self.chrom, self.pos, self.ref, self.alt)
