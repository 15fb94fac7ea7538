"""tests/oracle_cache.py: the hash-keyed fixtures of oracle outputs miss on ANY change of what the oracle's result depends on."""
import os
import re

import torch

import oracle_cache


def test_key_covers_weights_inputs_scalars_and_tag(tmp_path, monkeypatch):
    monkeypatch.setenv("ICD_ORACLE_CACHE_WRITE", str(tmp_path))
    monkeypatch.setattr(oracle_cache, "DIR", str(tmp_path))            # "committed" fixtures = what this test writes
    sd = {"w": torch.arange(12.0).reshape(3, 4), "b": torch.zeros(3)}
    x = torch.ones(2, 3)
    calls = []

    def compute():
        calls.append(1)
        return torch.full((2,), float(len(calls)))
    v1 = oracle_cache.lookup("case", sd, [x, [(999, 699)], None, 7.0], compute)
    v2 = oracle_cache.lookup("case", sd, [x.clone(), [(999, 699)], None, 7.0], compute)      # same key: the fixture, no recomputation
    assert len(calls) == 1 and torch.equal(v1, v2)
    variants = [
        ("case", {"w": sd["w"] + 1e-7 * 0, "b": sd["b"]}, [x, [(999, 699)], None, 7.0], False),         # identical bytes -> hit
        ("case", {"w": sd["w"].clone().index_put_((torch.tensor(0), torch.tensor(0)), torch.tensor(1e-3)), "b": sd["b"]},
         [x, [(999, 699)], None, 7.0], True),                                                         # one weight element
        ("case", sd, [x * 1.0000001, [(999, 699)], None, 7.0], True),                                  # one ulp in an input
        ("case", sd, [x, [(999, 698)], None, 7.0], True),                                              # the schedule
        ("case", sd, [x, [(999, 699)], None, 7.5], True),                                              # a scalar
        ("case", sd, [x.double(), [(999, 699)], None, 7.0], True),                                     # the dtype
        ("case", sd, [x.reshape(3, 2), [(999, 699)], None, 7.0], True),                                # the shape
        ("other", sd, [x, [(999, 699)], None, 7.0], True),                                             # the tag
    ]
    for tag, w, parts, recomputed in variants:
        n = len(calls)
        oracle_cache.lookup(tag, w, parts, compute)
        assert (len(calls) == n + 1) == recomputed, (tag, parts[1:], recomputed)
    monkeypatch.setenv("ICD_ORACLE_CACHE_OFF", "1")                      # the switch that forces the live oracle
    n = len(calls)
    oracle_cache.lookup("case", sd, [x, [(999, 699)], None, 7.0], compute)
    assert len(calls) == n + 1


def test_an_edit_to_the_oracle_sources_changes_every_key(tmp_path, monkeypatch):
    monkeypatch.setattr(oracle_cache, "DIR", str(tmp_path))
    monkeypatch.setenv("ICD_ORACLE_CACHE_WRITE", str(tmp_path))
    sd = {"w": torch.ones(2)}
    oracle_cache.lookup("case", sd, [1], lambda: torch.zeros(1))
    names = set(os.listdir(tmp_path))
    monkeypatch.setitem(oracle_cache._memo, "src", "an-edited-oracle")
    oracle_cache.lookup("case", sd, [1], lambda: torch.zeros(1))
    assert len(set(os.listdir(tmp_path)) - names) == 1


def test_committed_fixtures_are_well_formed():
    files = sorted(os.listdir(oracle_cache.DIR))
    assert files, "tests/golden/oracle_cache is empty"
    for f in files:
        assert re.fullmatch(r"[a-z0-9_]+-[0-9a-f]{32}\.pt", f), f
        t = torch.load(os.path.join(oracle_cache.DIR, f), map_location="cpu")
        assert isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] == 4 and torch.isfinite(t).all(), f
        assert os.path.getsize(os.path.join(oracle_cache.DIR, f)) < 1 << 20
