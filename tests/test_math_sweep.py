"""exp / log / pow / atan2 / sqrt of the path: device function == CPU restatement == (where a single target exists) torch.

tests/golden/math_sweep.json (oracle/gen_math_golden.py, build container) holds, per block of 2^24 inputs and
for all 2^32 inputs of each function, the checksum of oracle/rm_math_ref.c, the checksum of torch's CPU result on
the generating host, and how many inputs differ.  Here:
  * CPU: sample blocks of the restatement are recomputed against the fixture; pow and atan2 (Sleef inside
    ATen) must equal THIS host's torch bit for bit; exp and log (MKL VML inside ATen, host dependent) must
    stay within 1 ulp of it;
  * GPU: the device functions of ray_marching_amd/csrc/rm_math.h are run over all 2^32 inputs of every
    function and must reproduce all 4 x 256 checksums of the restatement.
"""
import ctypes as C
import json
import os

import pytest
import torch

from oracle import math_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "math_sweep.json")
SWEEP_LIB = os.path.join(os.path.dirname(__file__), "_build", "librm_math_sweep.so")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_fixture_is_exhaustive_and_sleef_functions_match_torch(gold):
    assert gold["block"] == 1 << 24
    for fn, rows in gold["functions"].items():
        assert len(rows["ours"]) == len(rows["torch"]) == len(rows["n_diff"]) == 256, fn
        assert rows["worst_ulp"] <= 1, (fn, rows["worst_ulp"])
    # Sleef inside ATen: one target, matched on every input
    for fn in ("pow_gamma", "atan2"):
        rows = gold["functions"][fn]
        assert rows["total_diff"] == 0 and rows["ours"] == rows["torch"], fn
    # MKL VML inside ATen: no single target; the recorded distance from the generating host's bits
    assert gold["functions"]["exp"]["total_diff"] < 0.0025 * 2 ** 32
    assert gold["functions"]["log"]["total_diff"] < 1e-5 * 2 ** 32
    assert gold["functions"]["sqrt"]["total_diff"] < 0.0035 * 2 ** 32       # vsSqrt(HA) is not correctly rounded


SAMPLE_BLOCKS = [0x00, 0x3f, 0x40, 0x42, 0x7f, 0x80, 0xbf, 0xc1, 0xff]


@pytest.mark.parametrize("fn", list(math_ref.FN))
def test_restatement_vs_fixture_and_this_hosts_torch(fn, gold):
    rows = gold["functions"][fn]
    for blk in SAMPLE_BLOCKS:
        a, b = math_ref.sweep_inputs(fn, blk)
        ref = math_ref.torch_eval(fn, a, b).contiguous()
        s, osum, nd, mu = math_ref.sweep_block(fn, blk, ref)
        assert f"{s:016x}" == rows["ours"][blk], (fn, blk)
        assert mu <= 1, (fn, blk, mu)
        if fn in ("pow_gamma", "atan2"):
            assert nd == 0, f"{fn} block {blk:#x}: {nd} inputs differ from this host's torch (Sleef: must be 0)"
        elif f"{osum:016x}" == rows["torch"][blk]:
            assert nd == rows["n_diff"][blk]        # same MKL code path as the generating host
        else:
            print(f"{fn} block {blk:#x}: this host's torch differs from the fixture host's ({nd} vs "
                  f"{rows['n_diff'][blk]} inputs away from the restatement) -- MKL VML is CPU-dispatched")


def test_restatement_special_values():
    inf, nan = float("inf"), float("nan")
    x = torch.tensor([0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 1e-40, 88.7, 88.8, -103.9, -104.1, 1e-45, 3e38])
    e = math_ref.expf(x)
    assert torch.equal(torch.isnan(e), torch.isnan(torch.exp(x)))
    for got, want in zip(e.tolist(), torch.exp(x.double()).float().tolist()):
        assert got == want or (got != got and want != want), (got, want)
    lg = math_ref.logf(x)
    for got, want in zip(lg.tolist(), torch.log(x.double()).float().tolist()):
        assert got == want or (got != got and want != want), (got, want)
    sp = torch.tensor([0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 1e-40, -1e-40, 3e38, 1e-45, 0.5, -2.0, 3.0])
    a, b = sp.repeat_interleave(len(sp)), sp.repeat(len(sp))
    # ATen's vectorised loop hands the last numel % 32 elements of every thread's chunk to the SCALAR libm
    # function (glibc atan2f / powf, not Sleef): pad to a multiple of 32 so the whole array takes the Sleef path.
    pad = -len(a) % 32
    a, b = torch.cat([a, torch.ones(pad)]), torch.cat([b, torch.ones(pad)])
    for ours, theirs in ((math_ref.atan2f(a, b), torch.atan2(a, b)), (math_ref.powf(a, b), torch.pow(a, b))):
        same = (ours.view(torch.int32) == theirs.view(torch.int32)) | (torch.isnan(ours) & torch.isnan(theirs))
        assert bool(same.all()), (a[~same], b[~same], ours[~same], theirs[~same])


# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sweep_lib():
    assert os.path.isfile(SWEEP_LIB), "tests/_build/librm_math_sweep.so missing: run __graft_entry__.build()"
    lib = C.CDLL(SWEEP_LIB)
    lib.rm_math_sweep.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
    lib.rm_math_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("fn", list(math_ref.FN))
def test_device_functions_reproduce_every_checksum(fn, gold, sweep_lib):
    """All 2^32 inputs through the device function: 256 checksums == the restatement's."""
    sums = torch.zeros(256, dtype=torch.int64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = sweep_lib.rm_math_sweep(math_ref.FN[fn], 0, 256, math_ref.GAMMA, C.c_void_p(sums.data_ptr()), stream)
    assert rc == 0
    torch.cuda.synchronize()
    got = [f"{v & 0xffffffffffffffff:016x}" for v in sums.cpu().tolist()]
    want = gold["functions"][fn]["ours"]
    ranges = gold["functions"][fn].get("checked_blocks", [[0, 256]])     # sqrt_rn is specified for |x| >= 2^-95
    bad = [i for lo, hi in ranges for i in range(lo, hi) if got[i] != want[i]]
    assert not bad, f"{fn}: device differs from the restatement in blocks {[hex(b) for b in bad[:16]]} ({len(bad)} of 256)"
    if fn in ("pow_gamma", "atan2"):
        assert got == gold["functions"][fn]["torch"]     # hence bit-identical with ATen (Sleef) on every input


@pytest.mark.gpu
def test_device_special_values(sweep_lib):
    inf, nan = float("inf"), float("nan")
    sp = torch.tensor([0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 1e-40, -1e-40, 3e38, 1e-45, 0.5, -2.0, 3.0, 88.75, -104.5])
    a, b = sp.repeat_interleave(len(sp)).contiguous(), sp.repeat(len(sp)).contiguous()
    ad, bd = a.cuda(), b.cuda()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for fn, ref in (("exp", math_ref.expf(a)), ("log", math_ref.logf(a)), ("pow_gamma", math_ref.powf(a, b)),
                    ("atan2", math_ref.atan2f(a, b)), ("sqrt", math_ref.sqrtf(a))):
        if fn == "sqrt":           # below 2^-95 the device function is not specified
            ref = torch.where(a.abs() < 2.0 ** -95, torch.zeros_like(ref), ref)
        out = torch.empty_like(ad)
        assert sweep_lib.rm_math_eval(math_ref.FN[fn], C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()),
                                      C.c_void_p(out.data_ptr()), ad.numel(), stream) == 0
        got = out.cpu()
        if fn == "sqrt":
            got = torch.where(a.abs() < 2.0 ** -95, torch.zeros_like(got), got)
        same = (got.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(got) & torch.isnan(ref))
        assert bool(same.all()), (fn, a[~same], b[~same], got[~same], ref[~same])
