"""oracle/libmaple_cpu.so -- the CPU twin of libmaple_hip.so ("a CPU build of the same .so exports the
identical ABI") -- driven through the SAME binding class (maple_amd.runtime.Device, handed the twin explicitly) by the SAME
test bodies the GPU parity suite runs against libmaple_hip.so (tests/test_hip_parity.py, tests/test_hip_search.py): the
reference's recorded calls of appendProbNode, mergeVectors (+ returnLK), estimateBranchLengthWithDerivative,
evaluatePlacement, rootVector, passGenomeListThroughBranch, shorten, areVectorsDifferent, the model tables, list updates,
and findBestParentTopology / the worker on the frozen trees -- so the two libraries can be swapped and diffed call for call.
Runs without a GPU; it also exercises the ctypes binding and the packed list format on the CPU.

The twin is test infrastructure (oracle/): nothing under maple_amd/ loads it."""
import ctypes as C
import os
import subprocess

import pytest

import test_hip_parity as P
import test_hip_search as S
from golden_util import load, model_args, ref_indices

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
TWIN_EXPORTS = ["maple_abi_version", "maple_create", "maple_destroy", "maple_last_error", "maple_set_fatal_policy", "maple_set_model",
                "maple_get_model", "maple_lists_upload", "maple_lists_update", "maple_lists_sizes", "maple_lists_download",
                "maple_arena_mark", "maple_arena_release", "maple_arena_stats", "maple_mutations_upload", "maple_append_batch",
                "maple_merge_batch", "maple_blen_batch", "maple_differ_batch", "maple_pass_branch_batch", "maple_shorten_batch",
                "maple_root_vector_batch", "maple_root_prob_batch", "maple_evaluate_placement_batch", "maple_tree_upload",
                "maple_spr_search_batch"]


@pytest.fixture(scope="module")
def twin():
    subprocess.check_call(["make", "-C", ORACLE, "-s"])
    lib = C.CDLL(os.path.join(ORACLE, "libmaple_cpu.so"))
    lib.maple_last_error.restype = C.c_char_p
    for name in TWIN_EXPORTS:
        getattr(lib, name)
    return lib


def twin_device(lib, ctx):
    from maple_amd.runtime import Device
    return Device(ref_indices(ctx), ctx["rootFreqs"], thresholdProb=ctx["thresholdProb"], minBLenSensitivity=ctx["minBLenSensitivity"],
                  thresholdDiffForUpdate=ctx["thresholdDiffForUpdate"], thresholdFoldChangeUpdate=ctx["thresholdFoldChangeUpdate"],
                  defaultBLen=ctx["defaultBLen"], lib=lib)


def test_twin_has_the_abi_version_of_the_header(twin):
    import re
    hdr = open(os.path.join(os.path.dirname(ORACLE), "include", "maple_hip.h")).read()
    assert twin.maple_abi_version() == int(re.search(r"#define MAPLE_ABI_VERSION (\d+)", hdr).group(1))


@pytest.mark.parametrize("name", P.FIXTURES)
def test_operator_parity_tests_pass_on_the_cpu_twin(twin, name):
    f = load(name)
    dev = twin_device(twin, f["context"])
    env = (f, dev, P.make_oracle(f))
    for body in (P.test_model_tables, P.test_appendProbNode, P.test_lists_update_keeps_ids_and_changes_contents, P.test_mergeVectors,
                 P.test_estimateBranchLength, P.test_evaluatePlacement, P.test_rootVector, P.test_structural_functions,
                 P.test_appendProbNode_log_of_zero_is_minus_infinity, P.test_ops_mirror_reads_like_the_reference):
        body(env)
    dev.close()


@pytest.mark.parametrize("name", [n for n in S.NAMES if "b1429" not in n][:4])
def test_spr_search_parity_tests_pass_on_the_cpu_twin(twin, name):
    from maple_amd.tree_host import HostTree
    f = S.load(name)
    ctx, t = f["context"], f["tree"]
    dev = twin_device(twin, ctx)
    dev.set_model(**model_args(f["model"]))
    tree = HostTree(t["root"], t["up"], t["children"], t["dist"], t["mutations"], t["nMinor"], t["probVect"], t["probVectUpRight"],
                    t["probVectUpLeft"], t["probVectTotUp"]).upload(dev)
    env = (f, dev, tree)
    S.test_spr_search_matches_reference(env)
    S.test_nodes_not_searched(env)
    dev.close()
