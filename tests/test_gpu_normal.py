"""GPU parity, normal (TSC) bursts: libtrxsig's HIP path through the C-ABI against (1) the golden
vectors captured from the real reference and (2) the CPU oracle on seeded random batches.
Everything is compared value-exact (IEEE ==): amplitude, TOA, soft bits, hard bits, flags."""
import os

import numpy as np
import pytest

import _pkg
import oraclebind
import synth
from util import GpuBatch, assert_veq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = {s: pkg.TrxSig(s, 0) for s in (1, 2, 4)}
    for v in c.values():
        v.use_torch_stream()
    return c


@pytest.mark.parametrize("sps", [1, 2, 4])
def test_device_tables_roundtrip(pkg, ctx, sps):
    assert np.array_equal(ctx[sps].tables_export(), pkg.build_tables_host(sps))


@pytest.mark.parametrize("name", ["normal_sps4.npz", "normal_sps1.npz"])
def test_golden_normal(pkg, ctx, golden, name):
    g = golden(name)
    sps = int(g["sps"]); t = ctx[sps]
    for tsc in range(8):
        sel = np.flatnonzero(g["tsc"] == tsc)
        gb = GpuBatch(g["x"], g["off"][sel], g["len"][sel], nsoft=148, stride=160)
        # energy gate disabled: the golden file holds analyzeTrafficBurst for every burst
        t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr,
                              hard=gb.hard, detect_thresh=3.0, energy_thresh=-1.0, nsoft=148, soft_stride=160)
        r = gb.results()
        det = (r["flags"] & pkg.F_DETECT) != 0
        assert_veq(det, g["ok"][sel].astype(bool), "detect flags tsc %d" % tsc)
        assert np.all(r["flags"] & pkg.F_ENERGY)
        assert_veq(r["amp"], g["amp"][sel], "amp"); assert_veq(r["toa"], g["toa"][sel], "toa")
        assert_veq(r["pwr"], g["energy_pwr"][sel], "energyDetect avgPwr")
        for j, i in enumerate(sel):
            if det[j]:
                assert_veq(r["soft"][j, :148], g["soft"][i, :148], "soft %d" % i)
                assert_veq(r["hard"][j, :148], (g["soft"][i, :148] > 0.5).astype(np.uint8), "hard %d" % i)
            else:
                assert not r["soft"][j, :148].any() and not r["hard"][j, :148].any()
            assert np.all(r["soft"][j, 148:] == -1.0)          # nothing written past nsoft
        # with the reference's start-up threshold (250) the energy gate follows energyDetect
        gb2 = GpuBatch(g["x"], g["off"][sel], g["len"][sel])
        t.detect_demod_normal(gb2.x, gb2.off, gb2.len, tsc, gb2.flags, gb2.amp, gb2.toa, gb2.soft,
                              energy_thresh=float(g["energy_thresh"]))
        r2 = gb2.results()
        e_ok = (r2["flags"] & pkg.F_ENERGY) != 0
        assert_veq(e_ok, g["energy_ok"][sel].astype(bool), "energy flags")
        assert_veq((r2["flags"] & pkg.F_DETECT) != 0, g["ok"][sel].astype(bool) & e_ok)
        assert not r2["amp"][~e_ok].any() and not r2["toa"][~e_ok].any()


@pytest.mark.parametrize("sps,B", [(4, 4096), (2, 1024), (1, 2048)])
def test_random_batch_vs_oracle(pkg, ctx, sps, B):
    o = oraclebind.Oracle(sps)
    for tsc in (0, 5):
        x, off, length, meta = synth.normal_batch(sps, B, tsc, seed=1234 + tsc + 10 * sps)
        gb = GpuBatch(x, off, length, nsoft=156, stride=157)
        ctx[sps].detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr,
                                     hard=gb.hard, energy_thresh=0.0, nsoft=156, soft_stride=157)
        r = gb.results()
        ok, amp, toa, soft = o.normal_batch(x, off, length, tsc, nsoft=156, nthreads=8)
        assert_veq((r["flags"] & pkg.F_DETECT) != 0, ok.astype(bool), "detect")
        assert_veq(r["amp"], amp, "amp"); assert_veq(r["toa"], toa, "toa")
        assert_veq(r["soft"][:, :156], soft, "soft")
        assert_veq(r["hard"][:, :156], (soft > 0.5).astype(np.uint8), "hard")
        # sanity on the physics: clean bursts demodulate to the transmitted bits
        clean = np.flatnonzero(ok.astype(bool) & (meta["sigma"] <= 0.1))
        assert len(clean) > B // 4
        assert np.array_equal(r["hard"][clean][:, :148], meta["bits"][clean])


@pytest.mark.parametrize("det_cus,layout", [(0, 0), (96, 0), (128, 1)])
def test_demodulator_beside_the_next_correlator(pkg, det_cus, layout):
    """(det_cus > 0: TRXSIG_TUNE_BESIDE_DET_CUS -- the detectors on a stream masked to that many compute units, the demodulator on
    the others.)  TRXSIG_TUNE_DEMOD_BESIDE: five different batches back to back, every call's demodulator on the context's side stream from
    its own copy of (flags, amp, TOA) while the next call's correlator runs; one trxsig_synchronize at the end.  Every output of
    every call equals the default mode's (two private copies alternate: the third call waits for the first one's demodulator)."""
    sps, tsc = 4, 3
    t = pkg.TrxSig(sps, 0); t.use_torch_stream()
    sizes = (3000, 4096, 1, 2500, 4096)
    batches = []
    for i, B in enumerate(sizes):
        x, off, length, _ = synth.normal_batch(sps, B, tsc, seed=900 + i, sigmas=(0.0, 0.1, 0.5, 2.0))
        batches.append((GpuBatch(x, off, length, nsoft=156, stride=157), GpuBatch(x, off, length, nsoft=156, stride=157)))
    for ref, _ in batches:
        t.detect_demod_normal(ref.x, ref.off, ref.len, tsc, ref.flags, ref.amp, ref.toa, ref.soft, avgpwr=ref.pwr, hard=ref.hard,
                              energy_thresh=0.0, nsoft=156, soft_stride=157)
    t.set_tuning(demod_beside=1, beside_det_cus=det_cus, cu_layout=layout)
    for _, gb in batches:
        t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                              energy_thresh=0.0, nsoft=156, soft_stride=157)
    t.synchronize()
    for ref, gb in batches:
        a, b = ref.results(), gb.results()
        for key in a:
            assert_veq(a[key], b[key], key)
    t.set_tuning(demod_beside=0, beside_det_cus=0)          # (a call of the ordinary kind afterwards is ordered behind everything)
    ref, gb = batches[1]
    gb.soft.fill_(-1.0)
    t.detect_demod_normal(gb.x, gb.off, gb.len, tsc, gb.flags, gb.amp, gb.toa, gb.soft, avgpwr=gb.pwr, hard=gb.hard,
                          energy_thresh=0.0, nsoft=156, soft_stride=157)
    assert_veq(ref.results()["soft"], gb.results()["soft"], "soft after switching back")
    t.close()


def test_ragged_and_bad_bursts(pkg, ctx):
    """offset/length edge cases: bad lengths are flagged and skipped, neighbours are unaffected."""
    sps = 4
    x, off, length, meta = synth.normal_batch(sps, 37, 3, seed=77)      # ragged tail: 37 = 2*16 + 5
    o = oraclebind.Oracle(sps)
    ok, amp, toa, soft = o.normal_batch(x, off, length, 3, nthreads=2)
    length2 = length.copy(); off2 = off.copy()
    length2[5] = 91 * sps          # too short for the midamble window
    length2[6] = 158 * sps         # longer than the rotation table
    length2[7] = 624 + 1           # not a multiple of sps
    off2[8] = -4                   # negative offset
    off2[9] = off2[9] + 1          # odd offset is legal (8-byte aligned path); shifts the burst by a sample
    gb = GpuBatch(np.concatenate([x, np.zeros(700, np.complex64)]), off2, length2)
    ctx[sps].detect_demod_normal(gb.x, gb.off, gb.len, 3, gb.flags, gb.amp, gb.toa, gb.soft)
    r = gb.results()
    bad = np.array([5, 6, 7, 8])
    assert np.all(r["flags"][bad] == pkg.F_BADLEN)
    assert not r["soft"][bad].any() and not r["amp"][bad].any()
    x9 = x[off[9] + 1:off[9] + 1 + length[9]]
    r9 = o.analyze_traffic(np.concatenate([x9, np.zeros(length[9] - len(x9), np.complex64)]), 3)
    assert r["amp"][9] == r9["amp"] and r["toa"][9] == r9["toa"]
    good = np.setdiff1d(np.arange(37), np.append(bad, 9))
    assert_veq((r["flags"][good] & pkg.F_DETECT) != 0, ok[good].astype(bool))
    assert_veq(r["amp"][good], amp[good]); assert_veq(r["toa"][good], toa[good])
    assert_veq(r["soft"][good], soft[good])
    # empty batch is a no-op
    ctx[sps].detect_demod_normal(gb.x, gb.off[:0], gb.len[:0], 3, gb.flags, gb.amp, gb.toa, gb.soft)


def test_demodulate_entry_point(pkg, ctx):
    """trxsig_demodulate_batch with caller-supplied amp/TOA (off the 1/512 grid too)."""
    sps = 4
    rng = np.random.default_rng(5)
    x, off, length, meta = synth.normal_batch(sps, 64, 1, seed=78)
    o = oraclebind.Oracle(sps)
    amp = (meta["amp"] * (1 + 0.01 * rng.standard_normal(64))).astype(np.complex64)
    toa = rng.uniform(-3, 3, 64).astype(np.float32)
    toa[:8] = np.round(toa[:8])                   # integer delays: no fractional filter
    toa[8:12] = np.float32(0.005)                 # |frac| below the 1e-2 gate
    toa[12:14] = np.float32(1e-9)                 # delay = -1e-9: floor gives -1 and the fraction rounds to exactly 1.0
    toa[14] = np.float32(-1e-9)
    en = np.ones(64, np.uint8); en[20] = 0
    import torch
    gb = GpuBatch(x, off, length, nsoft=156, stride=156)
    d_amp = torch.from_numpy(amp.view(np.float32).copy()).cuda()
    d_toa = torch.from_numpy(toa).cuda(); d_en = torch.from_numpy(en).cuda()
    ctx[sps].demodulate(gb.x, gb.off, gb.len, d_amp, d_toa, gb.soft, enable=d_en, hard=gb.hard, nsoft=156)
    r = gb.results()
    for i in range(64):
        ref = o.demodulate(x[off[i]:off[i] + length[i]], amp[i], toa[i])[:156]
        if en[i]:
            assert_veq(r["soft"][i], ref, "demod %d" % i)
        else:
            assert not r["soft"][i].any()


@pytest.mark.timeout(180)
def test_context_from_broadcast_tables(pkg):
    """The multi-GPU bench's context creation: the table blob built on the host, placed in device memory the way
    dist.broadcast_tables hands it over (here a single-rank RCCL group: init, broadcast and MAX all-reduce run, with
    nobody to talk to), validated, and trxsig_create_from_tables -- results identical to a context that built its
    own tables; a corrupted blob is refused."""
    import torch
    import torch.distributed as tdist
    from openbts_ttsou_amd import dist as d
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    own = not tdist.is_initialized()
    if own:
        tdist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        blob, t = d.broadcast_tables(pkg, 4, device=torch.device("cuda", 0), src=0)
        assert t.is_cuda and t.numel() == pkg.lib().trxsig_tables_bytes(4)
        assert d.max_over_ranks(1.25, torch.device("cuda", 0)) == 1.25
    finally:
        if own:
            tdist.destroy_process_group()
    a = pkg.TrxSig(4, 0, tables_blob=t); a.use_torch_stream()
    b = pkg.TrxSig(4, 0); b.use_torch_stream()
    x, off, length, meta = synth.normal_batch(4, 777, 3, seed=99)
    ra, rb = [], []
    for c, out in ((a, ra), (b, rb)):
        gb = GpuBatch(x, off, length)
        c.detect_demod_normal(gb.x, gb.off, gb.len, 3, gb.flags, gb.amp, gb.toa, gb.soft)
        out.append(gb.results())
    for k in ("flags", "amp", "toa", "soft"):
        assert_veq(ra[0][k], rb[0][k], k)
    bad = t.clone(); bad[5000] ^= 0x40
    with pytest.raises(pkg.TrxSigError):
        pkg.TrxSig(4, 0, tables_blob=bad)


def test_c_host_rccl_table_broadcast(tmp_path):
    """trxsig_tables_broadcast from a C++ host with no torch (tests/rccl_broadcast.cpp): a real ncclBroadcast on a one-rank
    communicator, then trxsig_create_from_tables on the received blob."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "openbts-ttsou_amd")
    exe = str(tmp_path / "rccl_broadcast")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tests", "rccl_broadcast.cpp"),
                           "-I", os.path.join(root, "include"), "-I/opt/rocm/include", "-L", lib, "-ltrxsig", "-L/opt/rocm/lib",
                           "-lrccl", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "broadcast ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])
