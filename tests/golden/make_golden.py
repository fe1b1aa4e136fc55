#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (BUILD CONTAINER ONLY).

Runs the unmodified reference (/root/reference/MAPLEv0.7.5.4.py, never copied,
never shipped) as ``__main__`` under ``sys.settrace`` and records, for the
hot-path functions of SURVEY.md §8(a), real call arguments, the model state the
call resolved, and the value the reference returned.  Calls are selected
greedily by *new reference source lines covered* (so rare branches are kept)
plus a seeded random reservoir, which keeps the fixtures small.

The output files hold DATA only (inputs / expected outputs); this script is
the committed recipe that made them.  Usage:

    python tests/golden/make_golden.py [run-name ...]
"""
import gzip
import json
import os
import random
import runpy
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/MAPLEv0.7.5.4.py"
EXAMPLE = "/root/reference/example_files/MAPLE_alignment_example.txt"

TARGETS = {
    "appendProbNode", "mergeVectors", "estimateBranchLengthWithDerivative",
    "areVectorsDifferent", "passGenomeListThroughBranch", "shorten", "rootVector",
    "getPartialVec", "simplify", "findProbRoot", "evaluatePlacement",
}
# per function: (max kept by coverage, reservoir size)
LIMITS = {
    "appendProbNode": (400, 250), "mergeVectors": (500, 300),
    "estimateBranchLengthWithDerivative": (300, 200), "areVectorsDifferent": (120, 80),
    "passGenomeListThroughBranch": (150, 100), "shorten": (60, 60), "rootVector": (80, 60),
    "getPartialVec": (60, 100), "simplify": (20, 40), "findProbRoot": (40, 30),
    "evaluatePlacement": (40, 60),
}


def ser_list(pv):
    """Genome list -> JSON-able nested lists (tuple lengths, bools and floats preserved)."""
    if pv is None:
        return None
    out = []
    for e in pv:
        out.append([list(x) if isinstance(x, (list, tuple)) else x for x in e])
    return out


def ser_ret(x):
    if isinstance(x, tuple):
        return [ser_ret(y) for y in x]
    if isinstance(x, list):
        if x and isinstance(x[0], tuple):
            return ser_list(x)
        return list(x)
    return x


class Harvester:
    def __init__(self, seed):
        self.rng = random.Random(seed)
        self.cov = {f: set() for f in TARGETS}
        self.kept_cov = {f: [] for f in TARGETS}
        self.reservoir = {f: [] for f in TARGETS}
        self.seen = {f: 0 for f in TARGETS}
        self.models = []
        self.model_keys = {}
        self.g = None
        self.active = {}

    # -- model snapshots -------------------------------------------------
    def model_id(self, g, loc):
        def pick(passed, glob):
            v = loc.get(passed)
            return v if v is not None else g.get(glob)
        Q = pick("mutMatrixGlobalPassed", "mutMatrixGlobal")
        mm = pick("mutMatricesGlobal", "mutMatrices")
        er = pick("errorRatesGlobal", "errorRates")
        eg = pick("errorRateGlobalPassed", "errorRateGlobal")
        te = pick("totErrorPassed", "totError")
        rv, ue, ss = bool(g["useRateVariation"]), bool(g["usingErrorRate"]), bool(g["errorRateSiteSpecific"])
        lRef = g["lRef"]
        key = (rv, ue, ss, tuple(tuple(r) for r in Q) if Q is not None else None, eg if ue else None,
               (id(mm), mm[0][0][1], mm[lRef // 2][1][3], mm[-1][3][2]) if (rv and mm is not None) else None,
               (id(er), er[0], er[lRef // 2], er[-1]) if (ue and ss and er is not None) else None,
               te if ue else None)
        mid = self.model_keys.get(key)
        if mid is not None:
            return mid
        snap = dict(useRateVariation=rv, usingErrorRate=ue, errorRateSiteSpecific=ss,
                    Q=[list(r) for r in Q] if Q is not None else None)
        if rv:
            sr = g["siteRates"]
            # the reference builds mutMatrices[i][j][k] = Q[j][k]*siteRates[i] (M:6350-6370): check, then store only siteRates
            for i in (0, lRef // 3, lRef - 1):
                for j in range(4):
                    for k in range(4):
                        assert mm[i][j][k] == Q[j][k] * sr[i], "mutMatrices not Q*siteRates"
            snap["siteRates"] = list(sr)
        if ue:
            snap["errorRateGlobal"] = eg
            snap["totError"] = te
            if ss:
                snap["errorRates"] = list(er)
        cr = loc.get("cumulativeRateGlobal")
        if cr is None:
            cr = g.get("cumulativeRate")
        snap["cumulativeRate_probe"] = [cr[1], cr[lRef // 2], cr[lRef]] if cr else None
        mid = len(self.models)
        self.models.append(snap)
        self.model_keys[key] = mid
        return mid

    # -- argument capture ------------------------------------------------
    def capture(self, name, loc, g):
        L = loc
        if name == "appendProbNode":
            a = dict(P=ser_list(L["probVectP"]), C=ser_list(L["probVectC"]), isTipC=bool(L["isTipC"]), bLen=L["bLen"])
        elif name == "mergeVectors":
            a = dict(pv1=ser_list(L["probVect1"]), b1=L["bLen1"], tip1=bool(L["fromTip1"]),
                     pv2=ser_list(L["probVect2"]), b2=L["bLen2"], tip2=bool(L["fromTip2"]),
                     returnLK=bool(L["returnLK"]), isUpDown=bool(L["isUpDown"]),
                     numMinor1=L["numMinor1"], numMinor2=L["numMinor2"])
        elif name == "estimateBranchLengthWithDerivative":
            a = dict(P=ser_list(L["probVectP"]), C=ser_list(L["probVectC"]), fromTipC=bool(L["fromTipC"]))
        elif name == "areVectorsDifferent":
            a = dict(pv1=ser_list(L["probVect1"]), pv2=ser_list(L["probVect2"]))
        elif name == "passGenomeListThroughBranch":
            a = dict(pv=ser_list(L["probVect"]), mutations=[list(m) for m in L["mutations"]], dirIsUp=bool(L["dirIsUp"]))
        elif name == "shorten":
            a = dict(vec=ser_list(L["vec"]))
        elif name == "rootVector":
            tree, node = L["tree"], L["node"]
            path = []
            n = node
            while n is not None:
                path.append([list(m) for m in tree.mutations[n]])
                n = tree.up[n]
            a = dict(pv=ser_list(L["probVect"]), bLen=L["bLen"], isFromTip=bool(L["isFromTip"]), pathMutations=path)
        elif name == "getPartialVec":
            a = dict(i12=L["i12"], totLen=L["totLen"], mutMatrix=[list(r) for r in L["mutMatrix"]],
                     errorRate=L["errorRate"], vect=list(L["vect"]) if L["vect"] is not None else None,
                     upNode=bool(L["upNode"]), flag=bool(L["flag"]))
        elif name == "simplify":
            a = dict(vec=list(L["vec"]), refA=L["refA"])
        elif name == "findProbRoot":
            path = []
            n = L["node"]
            while n is not None:
                path.append([list(m) for m in L["mutations"][n]])
                n = L["up"][n]
            a = dict(pv=ser_list(L["probVect"]), pathMutations=path)
        elif name == "evaluatePlacement":
            a = dict(midTot=ser_list(L["midTot"]), downVect=ser_list(L["downVect"]), upVect=ser_list(L["upVect"]),
                     distance=L["distance"], removedPartials=ser_list(L["removedPartials"]),
                     isRemovedTip=bool(L["isRemovedTip"]), fromTip1=bool(L["fromTip1"]))
        else:
            raise KeyError(name)
        if name not in ("getPartialVec", "simplify", "shorten", "passGenomeListThroughBranch", "areVectorsDifferent"):
            a["model"] = self.model_id(g, L)
        else:
            a["usingErrorRate"] = bool(g["usingErrorRate"])
        return a

    # -- tracing -----------------------------------------------------------
    def global_trace(self, frame, event, arg):
        if event != "call":
            return None
        co = frame.f_code
        if co.co_name not in TARGETS or not co.co_filename.endswith("MAPLEv0.7.5.4.py"):
            return None
        name = co.co_name
        if name in ("getPartialVec", "simplify") and self.seen[name] > 30000:
            return None
        try:
            rec = self.capture(name, frame.f_locals, frame.f_globals)
        except Exception as exc:  # capture must never break the run
            print("capture failed", name, repr(exc), file=sys.stderr)
            return None
        if self.g is None:
            self.g = frame.f_globals
        lines = set()
        vec_obj = frame.f_locals.get("vec") if name == "shorten" else None

        def local_trace(fr, ev, a):
            if ev == "line":
                lines.add(fr.f_lineno)
            elif ev == "return":
                rec["ret"] = ser_list(vec_obj) if name == "shorten" else ser_ret(a)
                self.finish(name, rec, lines)
            elif ev == "exception":
                rec["raised"] = True
            return local_trace
        return local_trace

    def finish(self, name, rec, lines):
        self.seen[name] += 1
        maxcov, rsize = LIMITS[name]
        new = lines - self.cov[name]
        if new and len(self.kept_cov[name]) < maxcov:
            self.cov[name] |= lines
            rec["why"] = "cov"
            self.kept_cov[name].append(json.dumps(rec))
            return
        res = self.reservoir[name]
        k = self.seen[name]
        if len(res) < rsize:
            res.append(json.dumps(rec))
        else:
            j = self.rng.randrange(k)
            if j < rsize:
                res[j] = json.dumps(rec)


def context_from_globals(g, ref_seq):
    keys = ["lRef", "thresholdProb", "minBLenSensitivity", "thresholdDiffForUpdate", "thresholdFoldChangeUpdate",
            "oneMutBLen", "effectivelyNon0BLen", "minimumCarryOver", "globalTotRate", "thresholdLogLK",
            "thresholdLogLKoptimization", "thresholdLogLKoptimizationTopology", "thresholdLogLKtopology",
            "thresholdLogLKconsecutivePlacement", "allowedFails", "allowedFailsTopology", "defaultBLen"]
    ctx = {k: g[k] for k in keys}
    ctx["rootFreqs"] = list(g["rootFreqs"])
    ctx["ref"] = ref_seq
    return ctx


def run_reference(name, input_file, flags, seed=7):
    out_dir = tempfile.mkdtemp(prefix="maple_golden_")
    out = os.path.join(out_dir, "out")
    argv = ["MAPLE", "--input", input_file, "--output", out, "--overwrite"] + flags
    h = Harvester(seed)
    old_argv = sys.argv
    sys.argv = argv
    t0 = time.time()
    crashed = None
    import io
    import contextlib
    log = io.StringIO()
    sys.settrace(h.global_trace)
    try:
        with contextlib.redirect_stdout(log):
            runpy.run_path(REF, run_name="__main__")
    except SystemExit:
        pass
    except Exception as exc:  # e.g. the reference raises for --model JC after round 1
        crashed = repr(exc)
    finally:
        sys.settrace(None)
        sys.argv = old_argv
    dt = time.time() - t0
    g = h.g
    ref_seq = g["ref"]
    calls = {}
    for f in sorted(TARGETS):
        items = [json.loads(s) for s in h.kept_cov[f]] + [json.loads(s) for s in h.reservoir[f]]
        calls[f] = items
    outputs = {}
    for suffix in ("_tree.tree", "_LK.txt", "_subs.txt", "_round1_preliminary_tree.tree"):
        p = out + suffix
        if os.path.exists(p):
            outputs[suffix] = open(p).read()
    lk_lines = [ln for ln in log.getvalue().splitlines() if "ikelihood" in ln or "LK" in ln][:200]
    fixture = dict(name=name, flags=flags, input=os.path.basename(input_file), crashed=crashed, wall_s=dt,
                   context=context_from_globals(g, ref_seq), models=h.models, calls=calls,
                   seen={k: v for k, v in h.seen.items()}, outputs=outputs, log_lk=lk_lines)
    path = os.path.join(HERE, f"calls_{name}.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(fixture, fh)
    print(f"[{name}] {dt:.1f}s crashed={crashed} models={len(h.models)} "
          + " ".join(f"{k}:{len(v)}/{h.seen[k]}" for k, v in calls.items())
          + f" -> {os.path.getsize(path)/1e6:.2f} MB", flush=True)


def synth_input(path, seed=11):
    from maple_amd.synth import make_dataset, write_maple
    d = make_dataset(n_samples=160, l_ref=1500, seed=seed, mean_diffs=14.0, rate_variation=True,
                     frac_with_n=0.35, n_run_len=(5, 120), frac_ambig=0.35, frac_ambig3=0.1)
    write_maple(d, path)


RUNS = {
    "example_unrest": (EXAMPLE, ["--model", "UNREST"]),
    "synth_unrest": ("SYNTH", ["--model", "UNREST", "--maxNumDescendantsForMATClade", "12"]),
    "synth_ratevar": ("SYNTH", ["--model", "UNREST", "--rateVariation", "--maxNumDescendantsForMATClade", "12"]),
    "synth_siteerr": ("SYNTH", ["--model", "UNREST", "--rateVariation", "--estimateSiteSpecificErrorRate",
                                "--maxNumDescendantsForMATClade", "12"]),
    "synth_gtr_err": ("SYNTH", ["--model", "GTR", "--estimateErrorRate", "--maxNumDescendantsForMATClade", "12"]),
    "synth_jc": ("SYNTH", ["--model", "JC", "--maxNumDescendantsForMATClade", "12"]),
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(RUNS)
    synth_path = os.path.join(HERE, "synth_small.maple.txt")
    if not os.path.exists(synth_path):
        synth_input(synth_path)
    for nm in which:
        inp, flags = RUNS[nm]
        run_reference(nm, synth_path if inp == "SYNTH" else inp, flags)
