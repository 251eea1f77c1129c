"""The composed drivers above the aligner (retarget / grid_search / update_read_info, the overhang filter,
find_by_smith_waterman_realn, the pileup front-end) against vectors produced by the reference's own function text
(oracle/gen_driver_golden.py -> tests/golden/driver_cases.json).  CPU: the host code is the package's, the alignments come
from the oracle through the GpuAligner interface (tests/port_backend.py); tests/test_gpu_driver.py replays the same vectors
through libindelpost_hip.so."""
import pytest

from tests import driver_replay as DR


@pytest.fixture()
def port_as_gpu(oracle_mod, monkeypatch):
    from indelpost_amd import localn, retarget, sswpy
    from tests.port_backend import PortAligner
    cache = {}

    def fake(device=0):
        if device not in cache:
            cache[device] = PortAligner(oracle_mod, device)
        return cache[device]
    for m in (sswpy, localn, retarget):
        monkeypatch.setattr(m, "_gpu", fake)
    return cache


@pytest.fixture(scope="module")
def cases():
    return DR.load()


def test_pileup_front_end_matches_the_reference(cases, port_as_gpu):
    """make_pileup / fetch_reads / dictize_read / get_ref_seq / get_local_reference (spliced and unspliced): every read dict,
    field by field (bulky fields by digest), and every per-read window"""
    n = sum(DR.replay(sc, cases["genomes"], parts=("pileup", "local_reference")) for sc in cases["scenarios"])
    assert n > 2500


def test_retarget_and_grid_search_match_the_reference(cases, port_as_gpu):
    """retarget under single penalty pairs (hits, misses, decoys, the window / 3 recursion) and the whole grid search with
    update_read_info on the winning reads: candidate, reads, gap penalties, every updated read dict"""
    n = 0
    for sc in cases["scenarios"]:
        n += DR.replay(sc, cases["genomes"], parts=("retarget", "grid_search"))
    assert n == sum(len(sc["retarget"]) + 1 for sc in cases["scenarios"])
    g = port_as_gpu[0]
    # the batching claim: one call per recursion level of a retarget / grid search, not one per read or per penalty pair
    assert g.n_calls < 1.5 * n and g.n_jobs > 10 * g.n_calls


def test_overhang_filter_and_smith_waterman_realn_match_the_reference(cases, port_as_gpu):
    n = sum(DR.replay(sc, cases["genomes"], parts=("overhangs", "realn", "parse", "perfect")) for sc in cases["scenarios"])
    assert n > 1000
    assert sum(1 for sc in cases["scenarios"] if sc["overhangs"]) >= 4


def test_many_loci_share_one_batch_per_level(cases, port_as_gpu):
    """grid_search_many / find_by_smith_waterman_realn_many: every scenario of the fixture as ONE request list -- the same
    answers as locus by locus, from a handful of aligner calls (one per recursion level for the searches, one for the realignment)"""
    from indelpost_amd import localn, pileup as P, retarget as RT, varaln
    from indelpost_amd.variant import Variant
    reqs, reqs2, exp = [], [], []
    for sc in cases["scenarios"]:
        genome = cases["genomes"][sc["genome"]]
        fa = DR.Fasta({"chr1": genome})
        chrom, pos, ref, alt = sc["target"]
        target = Variant(chrom, pos, ref, alt, fa)
        unspl = RT.UnsplicedLocalReference(chrom, pos, len(genome), sc["window"], fa)
        pile, _ = P.make_pileup(target, DR.Bam("chr1", [DR.Segment(r) for r in sc["segments"]]), unspl, sc["exclude_duplicates"], sc["window"],
                                sc["downsamplethresh"], sc["basequalthresh"])
        reqs.append((target, DR.clone(pile), sc["window"], 1, sc["within"], sc["cutoff"], 3, 2, [tuple(p) for p in sc["grid"]], unspl, False))
        pl = DR.clone(pile)
        for r in pl:
            r["is_target"] = r["read_name"] in sc["pre_target"]
        reqs2.append((target, DR.Contig(*sc["contig"]), pl, 3, 2, 3, 1, sc["basequalthresh"]))
        exp.append(sc)
    g = RT._gpu(0)                                                    # (the patched factory: the PortAligner of this test)
    c0 = g.n_calls
    out = varaln.grid_search_many(reqs)
    assert g.n_calls - c0 <= 4                                        # window, window / 3, / 9, / 27: one batch per level
    for res, sc in zip(out, exp):
        e = sc["grid_search"]
        if e is None:
            assert res is None
        else:
            assert DR.ser(res[0]) == e["candidate"] and (res[2], res[3]) == (e["gap_open"], e["gap_ext"])
            assert [DR.ser_read(r) for r in res[1]] == e["reads"]
    c0 = g.n_calls
    out2 = localn.find_by_smith_waterman_realn_many(reqs2)
    assert g.n_calls - c0 == 1
    for res, sc in zip(out2, exp):
        got = [{"read_name": r["read_name"], "is_target": r.get("is_target"), "undetermined": r.get("undetermined", False),
                "mismatches#": DR.dig(r["mismatches"])} for r in res]
        assert got == sc["realn"]


def test_check_overhangs_default_splice_rate_matches_the_reference():
    """pileup.pyx:435: the default threshold is 0.2 and the only caller passes none (varaln.pyx:260) -- loci whose dominant intron is
    supported by 10-20 % of the junctional reads must answer None.  Vectors: oracle/gen_overhang_rate_golden.py (reference text)."""
    import json
    import os
    from indelpost_amd import pileup as P
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "overhang_rate_cases.json")))["cases"]
    between = 0
    for c in cases:
        pl = [{"intron_pattern": tuple(r[0]), "is_covering": r[1], "covering_subread": tuple(r[2]) if r[2] else None, "aln_start": r[3], "aln_end": r[4]}
              for r in c["pileup"]]
        ans = P.check_overhangs(pl)
        got = None if ans is None else {"intron": list(ans[0]), "overhangs": [i for i, r in enumerate(pl) if any(r is o for o in ans[1])]}
        assert got == c["expected"], (c["support"], got, c["expected"])
        between += 0.1 <= c["support"] < 0.2
    assert between >= 20


def test_realignment_filters_match_the_reference_on_odd_cigars():
    """findall_mismatches / is_worth_realn in their pileup-level array forms (localn.findall_mismatches_pileup, worth_realn_mask) and
    as single-read calls against vectors from the reference's function text (oracle/gen_realn_filter_golden.py): 700 reads -- a soft clip
    behind a hard clip, an insertion right behind the clip, =/X/P tokens, introns, lower-case reference bases, aln_end one off --
    x two end trims, and 12 targets whose shiftable span reaches over read ends"""
    import copy
    import json
    import os
    from indelpost_amd import localn
    with open(os.path.join(os.path.dirname(__file__), "golden", "realn_filter_cases.json")) as f:
        fx = json.load(f)

    class Shift:
        def __init__(self, pos):
            self.pos = pos

    class Target:
        def __init__(self, t):
            self.pos, self.ref, self.alt, self.is_ins, self.n_made = t["pos"], t["ref"], t["alt"], len(t["alt"]) > len(t["ref"]), 0
            self._shifts = t["shifts"]

        def generate_equivalents(self):
            self.n_made += 1
            return [Shift(p) for p in self._shifts]

    def reads():
        out = copy.deepcopy([c["read"] for c in fx["reads"]])
        for r in out:
            r["covering_subread"] = tuple(r["covering_subread"]) if r["covering_subread"] else None
        return out

    as_lists = lambda ms: [list(m) for m in ms]
    for trim in (0, 3):
        pile = localn.findall_mismatches_pileup(reads(), trim)                 # the whole fixture as ONE pileup
        singles = reads()
        for r, one, case in zip(pile, singles, fx["reads"]):
            assert as_lists(r["mismatches"]) == case["mismatches"][str(trim)], r["cigar_string"]
            assert localn.findall_mismatches(one, trim) is one and one["mismatches"] == r["mismatches"]
            assert all(type(m[0]) is int and type(m[3]) is int for m in r["mismatches"])
    pile = localn.findall_mismatches_pileup(reads())
    n_true = 0
    for k, t in enumerate(fx["targets"]):
        tgt = Target(t)
        mask = localn.worth_realn_mask(pile, tgt, t["qual_lim"])
        assert tgt.n_made == 1                                                 # (the reference regenerates the equivalents per read)
        assert mask.tolist() == [c["worth"][k] for c in fx["reads"]]
        n_true += int(mask.sum())
        for j in range(k, len(pile), 12):
            assert localn.is_worth_realn(pile[j], tgt, t["qual_lim"]) is fx["reads"][j]["worth"][k]
    assert n_true > 1000
    # the plan of a pileup: reads that are targets already stay, reference reads and low mapping qualities are not asked
    for r, flag in zip(pile, range(len(pile))):
        r["is_target"], r["mapq"] = flag % 7 == 0, 0 if flag % 5 == 0 else 40
    tgt = Target(fx["targets"][0])
    plan = localn.realn_plan(pile, tgt, 1)
    worth = localn.worth_realn_mask(pile, tgt)
    for r, p, w in zip(pile, plan.tolist(), worth.tolist()):
        assert p == (-1 if r["is_target"] else int(not r["is_reference_seq"] and r["mapq"] > 1 and w))
