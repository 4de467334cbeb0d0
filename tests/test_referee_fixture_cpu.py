"""The committed referee traces (tests/golden/referee_traces.json, tests/referee_cache.py) are complete and well formed.  Whether an
entry still belongs to the window a GPU test builds is decided there, by hash; here only the shape of the file."""
import json
import os

import referee_cache as rc


def test_fixture_covers_every_run():
    assert os.path.exists(rc.FIXTURE), "run tests/golden/make_referee_traces.py"
    fx = json.load(open(rc.FIXTURE))
    assert set(fx) >= set(rc.RUNS)
    for key, (window, opts, _) in rc.RUNS.items():
        ent = fx[key]
        assert len(ent["hash"]) == 40 and len(ent["cams"]) == 8 and all(len(r) == 6 for r in ent["cams"])
        its = ent["iterations"]
        assert its[0]["iteration"] == 0 and all(set(rc.ITER_KEYS) <= set(i) for i in its)
        if "max_num_iterations" in opts and not key.endswith("referee") and "twin" not in key:
            assert len(its) == opts["max_num_iterations"] + 1, key        # fixed-length runs
        assert ent["final_cost"] <= ent["initial_cost"]
    # the hash covers the window, the oracle's sources AND the run's options: no two runs share one
    assert len({fx[k]["hash"] for k in rc.RUNS}) == len(rc.RUNS)
    assert fx["configs1/referee"]["termination_type"] == 0


def test_entry_hash_follows_options_and_oracle_sources(monkeypatch):
    """ an edited oracle or other solver options must not leave a fixture entry valid."""
    a = rc.options_fingerprint("configs1/referee")
    assert a != rc.options_fingerprint("configs1/twin_analytic") and a != rc.options_fingerprint("configs1/referee_2")
    assert "extended_precision" in a and "function_tolerance" in a
    before = rc.oracle_fingerprint()
    monkeypatch.setattr(rc, "_ORACLE_SOURCES", rc._ORACLE_SOURCES[:2])
    assert rc.oracle_fingerprint() != before
