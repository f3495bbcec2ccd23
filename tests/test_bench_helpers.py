"""CPU tests of bench.py's arithmetic: the SURVEY 8(d) byte formula against the worked values of BASELINE.md section 4,
the memory budget (cgroup headroom beats the tmpfs mount size) and the workload presets."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_survey_formula_matches_the_worked_values():
    # N = 2.6e8 -> 460 / 688 / 916 / 1144 / 1372 B for l = 1..5; cfg4 (ceil(log2 N) = 26): 428 / 640 / 852 / 1064 / 1276
    for n, want in ((260_000_000, (460, 688, 916, 1144, 1372)), (55_000_000, (428, 640, 852, 1064, 1276)), (100_000_000, (444, 664, 884, 1104, 1324))):
        for l in range(1, 6):
            lm = np.array([l], np.int32)                                  # one token matched to length l: lookups l = 1..l
            total, lookups = bench.survey_bytes(n, lm)
            assert lookups == l and total == sum(want[:l]), (n, l)


def test_memory_budget_is_bounded_by_what_the_process_may_use():
    budget, avail, cgroup = bench.memory_budget()
    assert budget > 0 and (avail is None or budget <= avail) and (cgroup is None or cgroup <= 0 or budget <= cgroup)
    assert bench.usable_cpus() >= 1


def test_presets_are_the_baseline_configs():
    c = bench.CONFIGS
    assert c["cfg3"]["pairs"] == 10_000_000 and c["cfg3"]["queries"] == 10_000 and c["cfg3"]["scaling"] == "weak"
    assert c["toy"]["pairs"] == 20_000 and c["toy"]["queries"] == 7                                 # the toy line (hansards stand-in)
    assert c["cfg4"]["queries"] == 50_000 and c["cfg4"]["scaling"] == "strong"
    assert c["cfg5"]["queries"] == 1_000_000 and abs(c["cfg5"]["pairs"] * 26 - 1e8) < 2e6          # 26 tokens per sentence pair incl. the delimiter


def test_chunks_respect_the_spool_and_the_internal_batch():
    """A step is cut into chunks of at most `max_queries` sentences and at most `tok_cap` query tokens; every sentence is
    in exactly one chunk, in order; an over-long sentence still gets a chunk."""
    import numpy as np
    import bench
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 40, 1000); lens[17] = 500
    qoff = np.concatenate([[0], np.cumsum(lens)[:-1]]); ntok = int(lens.sum())
    for max_q, cap in ((1000, 10**9), (64, 10**9), (1000, 300), (7, 120)):
        ch = bench.make_chunks(qoff, ntok, max_q, cap)
        assert ch[0][0] == 0 and ch[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        for a, b in ch:
            assert 0 < b - a <= max_q
            toks = int(lens[a:b].sum())
            assert toks <= cap or b - a == 1
    assert bench.make_chunks(np.zeros(0, np.int64), 0, 10) == [(0, 0)]
