"""Row N4: the two top-k candidate policies (Cerebro.cpp:366-492 naive, :506-722 clique).  The C restatement
(oracle/policies.c) is checked against an independent Python transcription of the same reference lines; both draw their
top-5 from the oracle's scan, so this pins the POLICY logic (state machine, locality rules, flush/thinning)."""
import numpy as np

import oracle_lib
import scenarios


def search5(db, ntotal, row):
    s, i = oracle_lib.scan_topk(db, ntotal, db[row:row + 1], 5)
    return s[0].astype(np.float32), i[0]


class PyNaive:
    def __init__(self, db):
        self.db, self.last_l, self.added = db, 0, 0

    def tick(self, l):
        if l - self.last_l < 3:
            return []
        if l > 150:
            self.added = l - 150
        tmp, tmp_i = [], []
        for l_i in range(self.last_l, l):
            if self.added < 5:
                continue
            d, lab = search5(self.db, self.added, l_i)
            tmp.append(d[0]); tmp_i.append(int(lab[0]))
        out = []
        if len(tmp) == 3 and tmp[2] > np.float32(0.9) and abs(tmp_i[0] - tmp_i[1]) < 12 and abs(tmp_i[0] - tmp_i[2]) < 12:
            out.append((l - 1, tmp_i[2], float(tmp[2])))
        self.last_l = l
        return out


class PyClique:
    def __init__(self, db, rnd):
        self.db, self.last_l, self.added, self.retained, self.rnd = db, 0, 0, {}, rnd

    def tick(self, l):
        if l <= self.last_l:
            return []
        if l > 150:
            self.added = l - 150
        out = []
        for l_i in range(self.last_l, l):
            if self.added < 5:
                break
            d, lab = search5(self.db, self.added, l_i)
            for g in range(5):
                if float(d[g]) < 0.85:
                    break
                dup = next((k for k in sorted(self.retained) if k - int(lab[g]) < 7), None)
                if dup is not None:
                    self.retained[dup] += 1
                else:
                    self.retained[int(lab[g])] = 1
            if self.retained and l_i % 4 == 0:
                keys = sorted(self.retained)
                if len(keys) == 1:
                    out.append((l - 1, keys[0], 0.9))
                else:
                    percent = int(100.0 / len(keys))
                    out += [(l - 1, k, 0.9) for k in keys if self.rnd() % 100 < percent]
                self.retained = {}
        self.last_l = l
        return out


def policy_db(seed=5, N=900, D=256):
    """Revisits: runs of 8 consecutive rows copying an earlier run (>= 150 + margin older), two of them."""
    plants = []
    for q0, p0 in ((400, 60), (700, 300), (820, 90)):
        plants += [(q0 + j, p0 + j, 1) for j in range(8)]
    return scenarios.build_db(seed, N, D, sorted(plants)), plants


def test_naive_policy_matches_transcription_and_finds_planted_revisits():
    db, plants = policy_db()
    ticks = list(range(3, db.shape[0] + 1, 3))
    orc, py = oracle_lib.NaivePolicyOracle(db), PyNaive(db)
    found = []
    for l in ticks:
        a, b = orc.tick(l), py.tick(l)
        assert a == b, (l, a, b)
        found += a
    src = {d: s for d, s, _ in plants}
    assert len(found) >= 3
    for cur, prev, score in found:
        assert abs(prev - src[cur]) < 12 and score > 0.9 and score == float(np.float32(score))
    # irregular schedule: ticks with != 3 new rows can never fire (the _n == 3 rule, :476)
    orc2 = oracle_lib.NaivePolicyOracle(db)
    assert all(orc2.tick(l) == [] for l in range(4, db.shape[0] + 1, 4))
    assert oracle_lib.NaivePolicyOracle(db).tick(2) == []                 # fewer than 3 new: nothing, state unchanged


def test_clique_policy_matches_transcription():
    db, plants = policy_db(seed=9)
    for step in (1, 2, 3, 5):
        orc = oracle_lib.CliquePolicyOracle(db, oracle_lib.AnsiRand())
        py = PyClique(db, oracle_lib.AnsiRand())
        total = 0
        for l in range(step, db.shape[0] + 1, step):
            a, b = orc.tick(l), py.tick(l)
            assert a == b, (step, l, a, b)
            total += len(a)
        assert total >= 3, step
        assert orc.st.n_retained == len(py.retained)


def test_clique_signed_locality_quirk_is_preserved():
    # (key - label) < 7 without abs() (Cerebro.cpp:634): a label far ABOVE an existing key still counts as that key's
    # duplicate.  Row 403 revisits row 63; with key 10 already retained, 63 is folded into key 10 instead of a new key.
    db, _ = policy_db(seed=9)
    orc = oracle_lib.CliquePolicyOracle(db)
    orc.st.last_l, orc.st.l_last_added, orc.st.n_retained = 403, 253, 1
    orc.st.key[0], orc.st.cnt[0] = 10, 1
    assert orc.tick(404) == []                       # 403 % 4 != 0: no flush
    assert orc.st.n_retained == 1 and orc.st.key[0] == 10 and orc.st.cnt[0] >= 2
    # ... whereas a key ABOVE label + 7 is not a duplicate: a new (smaller) key is inserted in front
    orc = oracle_lib.CliquePolicyOracle(db)
    orc.st.last_l, orc.st.l_last_added, orc.st.n_retained = 403, 253, 1
    orc.st.key[0], orc.st.cnt[0] = 200, 1
    orc.tick(404)
    assert orc.st.n_retained == 2 and abs(orc.st.key[0] - 63) < 7 and orc.st.key[1] == 200
