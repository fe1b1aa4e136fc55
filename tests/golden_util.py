"""Helpers shared by the parity tests: load the committed golden call records."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NUC = {"a": 0, "c": 1, "g": 2, "t": 3}


def fixture_names():
    return sorted(f[len("calls_"):-len(".json.gz")] for f in os.listdir(GOLDEN) if f.startswith("calls_"))


_cache = {}


def load(name):
    if name not in _cache:
        with gzip.open(os.path.join(GOLDEN, f"calls_{name}.json.gz"), "rt") as fh:
            _cache[name] = json.load(fh)
    return _cache[name]


def tup(gl):
    """JSON nested lists -> the reference's list of tuples (vectors stay lists)."""
    if gl is None:
        return None
    return [tuple(e) for e in gl]


def ref_indices(ctx):
    return np.array([NUC.get(ch, 0) for ch in ctx["ref"]], dtype=np.uint8)


def model_args(mod):
    """kwargs for set_model() from a model snapshot of the fixture."""
    return dict(Q=mod["Q"], siteRates=mod.get("siteRates") if mod["useRateVariation"] else None,
                usingErrorRate=mod["usingErrorRate"], errorRateGlobal=mod.get("errorRateGlobal", 0.0) or 0.0,
                errorRates=mod.get("errorRates") if mod["errorRateSiteSpecific"] else None)


def close(a, b, rel=1e-9, abs_=1e-300):
    if a == b:
        return True
    if a is None or b is None:
        return False
    return abs(a - b) <= max(abs_, rel * max(abs(a), abs(b)))


def lists_match(got, want, rel=1e-9):
    """Integer structure bit-exact, floats within rel."""
    if got is None or want is None:
        return got is None and want is None
    if len(got) != len(want):
        return False
    for g, w in zip(got, want):
        if len(g) != len(w):
            return False
        for x, y in zip(g, w):
            if isinstance(y, (list, tuple)):
                if len(x) != len(y) or not all(close(p, q, rel, 1e-18) for p, q in zip(x, y)):
                    return False
            elif isinstance(y, bool) or isinstance(x, bool):
                if bool(x) != bool(y):
                    return False
            elif isinstance(y, int):
                if x != y:
                    return False
            else:
                if not close(x, y, rel, 1e-18):
                    return False
    return True
