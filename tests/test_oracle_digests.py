"""The CPU oracle on seeded inputs against the committed digests (tests/golden/oracle_digests.json, written by
tests/golden/make_oracle_digests.py): pins the oracle itself -- the checker of every GPU parity test -- against
accidental change.  Not a statement about the reference (parity unpinned by the reference)."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_outputs_match_committed_digests():
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(HERE, "golden", "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))
    got = mod.compute()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k
