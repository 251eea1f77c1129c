"""Generate tests/golden/*.json from the reference itself (build container only).

Sources of truth used here, both built from /root/reference by oracle/Makefile and
oracle/build_ref_sswpy.py into oracle/_ref/ (git-ignored):
  * libssw_ref.so            -- the reference's ssw.c, compiled unmodified  -> C-level vectors
  * ref_sswpy_pkg.sswpy.SSW  -- the reference's Cython binding              -> Alignment tuples
Only DATA (inputs + expected outputs) is written to tests/golden/.  Run:  python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "_ref"))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LET = "ACGTN"


def s(codes):
    return "".join(LET[int(c)] for c in codes)


def c_level_cases(rng, n):
    ref = O.Backend("reference")
    scor = [(3, 2), (2, 2), (1, 1), (1, 3), (2, 4), (5, 4)]
    gaps = [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0), (1, 1), (1, 0), (0, 0), (0, 1), (1, 2), (2, 2), (6, 3), (10, 1)]
    cases = []
    for it in range(n):
        mode = it % 4
        wl = int(rng.integers(6, 420))
        rl = int(rng.integers(1, 256))
        alpha = 2 if mode == 2 else 4
        w = rng.integers(0, alpha, wl).astype(np.int8)
        if rng.random() < 0.85 and wl > 2:
            st = int(rng.integers(0, wl - 1))
            err = float(rng.choice([0.0, 0.02, 0.08, 0.2]))
            out, q = [], st
            while len(out) < rl:
                if q >= wl:
                    out.append(rng.integers(0, alpha)); continue
                u = rng.random()
                if u < err: out.append(rng.integers(0, alpha)); q += 1
                elif u < err * 1.5: q += 1
                elif u < err * 2: out.append(rng.integers(0, alpha))
                else: out.append(w[q]); q += 1
            r = np.array(out, np.int8)
        else:
            r = rng.integers(0, alpha, rl).astype(np.int8)
        if rng.random() < 0.2:
            r[rng.integers(0, len(r), max(1, len(r) // 10))] = 4
        if rng.random() < 0.1:
            w[rng.integers(0, len(w), max(1, len(w) // 10))] = 4
        ms, mm = scor[int(rng.integers(0, len(scor)))]
        go, ge = gaps[int(rng.integers(0, len(gaps)))]
        if rng.random() < 0.1:
            go = len(r)
        exp = ref.align(r, w, O.dna_matrix(ms, mm), go, ge)
        cases.append(dict(read=s(r), ref=s(w), match=ms, mismatch=mm, gap_open=int(go), gap_ext=int(ge), expect=exp))
    return cases


# inputs that once exposed a divergence of the GPU path (kept as permanent regression vectors)
REGRESSIONS = [
    # reverse 8-bit pass overflows (reverse score > forward score): the reference keeps the saturated
    # maximum but the previous column (ssw.c:325-331), so read_begin1 falls back to 0
    dict(read="CNNAACACACACAANACCAACAAAAACACCCCCAACCAANCCAACACAACAAAAANACCCCCNAACNCCACAACACAACCACCCAACCCACCCANACCCNCCANCCCCNNAACCACAACCACNAAAAACCANACAAAACNNNAAACAACCAACANAACCACAACCNACCCNACCCCACANCAACACA",
         ref="ACCCAAACCAACCCACAAACCCCACCACACCCCACCAAAACCCAAACACACAAACAAACACAAACACAC",
         match=5, mismatch=4, gap_open=2, gap_ext=2),
]


def regression_cases():
    ref = O.Backend("reference")
    out = []
    for c in REGRESSIONS:
        e = ref.align(O.encode(c["read"]), O.encode(c["ref"]), O.dna_matrix(c["match"], c["mismatch"]),
                      c["gap_open"], c["gap_ext"])
        out.append(dict(c, expect=e))
    return out


def sswpy_cases(rng):
    from ref_sswpy_pkg.sswpy import SSW
    out = []

    def add(ms, mm, ref, read, **kw):
        a = SSW(ms, mm)
        a.setReference(ref)
        a.setRead(read)
        r = a.align(**kw)
        out.append(dict(match=ms, mismatch=mm, ref=ref, read=read, kwargs=kw, expect=list(r)))

    R = "ACGTACGTTTGACCAGT"
    for rd in ["NNNN", "A", "ACGTACGT", "acgtacgt", "TTTT", "GGGGGGGG", "ACGTAGTTTGACCAGT", "ACGTACGTCCCTTGACCAGT", "UUGACC", ""]:
        add(3, 2, R, rd, gap_open=3, gap_extension=1)
    add(3, 2, R, "ACGTAGTTTGACCAGT", gap_open=16, gap_extension=1)
    add(3, 2, R, "ACGTAGTTTGACCAGT", gap_open=3, gap_extension=1, start_idx=4, end_idx=12)
    adv_read = "CCATGCTCACTCCAACCCGGCCCTGAGTCCGAGGAGAGGGGGCTTCAGAGTATTGGGTATGTACCTGGACTGGCA"
    adv_ref = "ATCACAGTCTACACTGCTCACTCCAACCCCGGCCCCTGAGTCCGAGGAGAGGGTGCTTCAGAGTATGTATACCACTGGGTAGGATACGGCGGAGGGCACGTCAATACGGTTCAATGCCCT"
    add(1, 3, adv_ref, adv_read, gap_open=1, gap_extension=1)
    for go, ge in [(3, 1), (3, 0), (5, 1), (5, 0), (4, 1), (4, 0), (75, 1)]:
        add(3, 2, adv_ref, adv_read, gap_open=go, gap_extension=ge)
    # random string-level cases incl. lower case, U, N, IUPAC junk, index windows, big penalties
    alpha = "ACGTacgtNnUuRY-"
    for it in range(120):
        wl = int(rng.integers(20, 300)); rl = int(rng.integers(1, 200))
        w = "".join(rng.choice(list("ACGT"), wl))
        st = int(rng.integers(0, max(1, wl - 5)))
        rd = list(w[st:st + rl]) + list(rng.choice(list("ACGT"), max(0, rl - (wl - st))))
        for k in range(len(rd)):
            u = rng.random()
            if u < 0.04: rd[k] = str(rng.choice(list(alpha)))
            elif u < 0.05: rd[k] = ""
            elif u < 0.06: rd[k] = rd[k] + str(rng.choice(list("ACGT")))
        rd = "".join(rd) or "A"
        if it % 3 == 0: w = w.lower()
        kw = dict(gap_open=int(rng.choice([3, 5, 4, 1, 0, len(rd), 300])), gap_extension=int(rng.choice([1, 0, 2])))
        if it % 5 == 0:
            a0 = int(rng.integers(0, wl // 2)); a1 = int(rng.integers(a0 + 1, wl + 1))
            kw.update(start_idx=a0, end_idx=a1)
        ms, mm = [(3, 2), (2, 2), (1, 1), (2, 4), (200, 3)][it % 5]
        add(ms, mm, w, rd, **kw)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    with open(os.path.join(OUT, "c_level_cases.json"), "w") as f:
        json.dump(dict(source="oracle/_ref/libssw_ref.so = /root/reference/indelpost/ssw.c compiled unmodified "
                              "(ssw_init(...,5,2) + ssw_align(flag=1, maskLen=max(15,len//2)))",
                       alphabet=LET, cases=regression_cases() + c_level_cases(rng, 480)), f, separators=(",", ":"))
    with open(os.path.join(OUT, "sswpy_cases.json"), "w") as f:
        json.dump(dict(source="reference indelpost/sswpy.pyx cythonized in the build container; "
                              "expect = list(Alignment) from SSW(match,mismatch).setReference/setRead/align(**kwargs)",
                       cases=sswpy_cases(rng)), f, separators=(",", ":"))
    # dataset checksums of SURVEY.md 8c/8d, re-measured with the reference library on the generator
    from indelpost_amd import synth
    jobs = synth.config2_jobs(100000)
    ref = O.Backend("reference")
    chk = {}
    for ms, mm in [(3, 2), (1, 1), (2, 2)]:
        _, c, ops = ref.cpu_baseline(jobs.reads, jobs.read_off, jobs.refs, jobs.ref_off, jobs.ref_id, jobs.gap_open,
                                     jobs.gap_ext, O.dna_matrix(ms, mm), 8)
        chk["%d,%d,3,1" % (ms, mm)] = dict(sum_score1=int(c), sum_cigar_len=int(ops))
    with open(os.path.join(OUT, "dataset_checksums.json"), "w") as f:
        json.dump(dict(source="SURVEY.md 8d generator, N=100000 x 150 bp vs one 300 bp window, reference library",
                       n=100000, checksums=chk), f, indent=1)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
