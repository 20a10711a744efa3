"""The Transceiver group (include/trxsig_trxgroup.h): pullRadioVector for S ARFCNs x n_slots timeslots per call, the
per-ARFCN state machine (adaptive energy threshold, false-detection clock, per-timeslot channel / DFE cache) replayed on
the device.  Checked against (1) S independent single-burst objects (include/trxsig_transceiver.h) fed the same bursts
one at a time and (2) oracle/transceiver_model.py on the CPU oracle: what comes back, its 148 soft bits, RSSI, timing
offset and the threshold after every burst (exact double equality).  S = 128 ARFCNs, 200 frames, channel combinations
I / II / IV / V / VII / NONE, four training sequences, clean bursts, two-path channels, loud noise (false detections raise
the threshold), silence (the threshold decays), bursts of the wrong kind; the frames arrive in calls of 1 ... 400 slots
that start on any timeslot."""
import numpy as np
import pytest

import _pkg
import oraclebind
import synth
import transceiver_model as tm

pytestmark = pytest.mark.gpu

CELL_SYM = 160                                   # samples per (slot, ARFCN) cell, in symbols (bursts are 156 / 157)


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _pkg.load()


def slot_config(a):
    """(tsc, {tn: combination}) of ARFCN a: the C0-like carriers carry the common channels, the rest traffic."""
    tsc = (2, 5, 0, 7)[a % 4]
    kind = a % 8
    if kind == 0:
        return tsc, {0: tm.V, 1: tm.VII, 2: tm.I, 3: tm.IV, 5: tm.II}
    if kind == 1:
        return tsc, {0: tm.IV, 2: tm.I, 4: tm.I, 6: tm.VII}
    if kind == 2:
        return tsc, {t: tm.I for t in range(8)}
    if kind == 3:
        return tsc, {1: tm.I, 2: tm.II, 3: tm.I, 7: tm.VI}
    if kind == 4:
        return tsc, {0: tm.V, 4: tm.I}
    if kind == 5:
        return tsc, {t: tm.I for t in (1, 3, 5, 7)}
    if kind == 6:
        return tsc, {0: tm.VII, 1: tm.VII, 2: tm.I}
    return tsc, {2: tm.I, 3: tm.I, 4: tm.V}


def configure(ctl, a):
    tsc, slots = slot_config(a)
    r = [ctl("CMD RXTUNE 890000"), ctl("CMD TXTUNE 935000"), ctl("CMD SETTSC %d" % tsc)]
    r += [ctl("CMD SETSLOT %d %d" % (tn, c)) for tn, c in sorted(slots.items())]
    r.append(ctl("CMD POWERON"))
    return r


def build_cells(sps, S, n_slots, fn0, tn0, seed, quiet_slots=(700, 1200)):
    """The received bursts, [n_slots][S][CELL] complex64, and each cell's expected correlation type."""
    rng = np.random.default_rng(seed)
    cell = CELL_SYM * sps
    x = np.zeros((n_slots, S, cell), np.complex64)
    pool_n, pool_r = 1536, 768
    pools = {}
    for tsc in (2, 5, 0, 7):
        xs, offs, lens, _ = synth.normal_batch(sps, pool_n, tsc, seed=seed + tsc, sigmas=(0.0, 0.05, 0.2, 0.5), max_delay=1.2)
        for i in range(0, pool_n, 3):                                    # a two-path channel on a third of them
            s = xs[offs[i]:offs[i] + lens[i]]
            s[sps:] = s[sps:] + np.complex64(0.35 - 0.2j) * s[:-sps].copy()
        pools[tsc] = (xs, offs, lens)
    pools["rach"] = synth.rach_batch(sps, pool_r, seed=seed + 11, sigmas=(0.0, 0.1, 0.3), max_delay_sym=20)[:3]
    ctype = np.zeros((n_slots, S), np.int8)
    chan = np.zeros((S, 8), np.int64)
    tscs = np.zeros(S, np.int64)
    for a in range(S):
        tscs[a], slots = slot_config(a)
        for tn, c in slots.items():
            chan[a, tn] = c
    m = tm.TransceiverModel.__new__(tm.TransceiverModel)
    for t in range(n_slots):
        tn = (tn0 + t) % 8
        fn = (fn0 + (tn0 + t) // 8) % tm.HYPERFRAME
        n = (156 + (tn % 4 == 0)) * sps
        quiet = quiet_slots[0] <= t < quiet_slots[1]                     # > 50 frames of silence: the thresholds decay
        for a in range(S):
            m.chan_type = chan[a]
            ct = m.expected_corr_type(tn, fn)
            ctype[t, a] = ct
            kind = rng.integers(0, 12)
            if quiet or kind == 0 or (ct in (tm.OFF, tm.IDLE) and kind < 8):
                v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.5
            elif kind <= 2:
                v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rng.uniform(300, 2500)     # loud noise
            else:
                want_rach = (ct == tm.RACH) != (kind == 3)               # kind 3: a burst of the other sort
                xs, offs, lens = pools["rach"] if want_rach else pools[int(tscs[a])]
                i = rng.integers(0, len(offs))
                v = xs[offs[i]:offs[i] + lens[i]][:n]
                if len(v) < n:
                    v = np.concatenate([v, np.zeros(n - len(v), np.complex64)])
            x[t, a, :n] = v
    return x, ctype


def run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls, pipelined=False, beside_rows=0):
    import torch
    ctx = pkg.TrxSig(sps, 0)
    ctx.use_torch_stream()
    g = pkg.TrxGroup(ctx, S, tsc_leg=leg, start=(fn0, tn0))
    if beside_rows:
        g.set_beside_rows(beside_rows)
    if pipelined:
        g.set_pipelined(True)
    cell = x.shape[2]
    out = dict(valid=np.zeros((n_slots, S), bool), soft=np.zeros((n_slots, S, 148), np.float32), rssi=np.zeros((n_slots, S), np.int32),
               timing=np.zeros((n_slots, S), np.int32), threshold=np.zeros((n_slots, S)))
    responses = [configure(lambda c, a=a: g.control(a, c), a) for a in range(S)]
    dx = torch.from_numpy(x.view(np.float32).reshape(-1)).to("cuda:0")
    t = 0
    k = 0
    while t < n_slots:
        n = min(calls[k % len(calls)], n_slots - t)
        k += 1
        tn = (tn0 + t) % 8
        fn = (fn0 + (tn0 + t) // 8) % tm.HYPERFRAME
        res = g.pull(dx.data_ptr() + 8 * t * S * cell, S * cell, cell, fn, tn, n)
        assert res.n_slots == n and res.n_arfcn == S
        r = g.collect()
        for key in out:
            out[key][t:t + n] = r[key]
        t += n
    final_thr = np.array([g.energy_threshold(a) for a in range(S)])
    g.close(); ctx.close()
    return out, responses, final_thr


def check_against(name, pull, ctype, x, sps, out, arfcns, fn0, tn0, thr_of):
    """pull(a, burst, tn, fn) -> None | (soft, rssi, timing); thr_of(a) -> the threshold after the burst."""
    n_slots = x.shape[0]
    seen = {"none": 0, "tsc": 0, "rach": 0}
    for a in arfcns:
        for t in range(n_slots):
            ct = ctype[t, a]
            if ct in (tm.OFF, tm.IDLE):
                assert not out["valid"][t, a] and np.isnan(out["threshold"][t, a])
                continue
            tn = (tn0 + t) % 8
            fn = (fn0 + (tn0 + t) // 8) % tm.HYPERFRAME
            n = (156 + (tn % 4 == 0)) * sps
            r = pull(a, x[t, a, :n], tn, fn)
            assert out["threshold"][t, a] == thr_of(a), (name, a, t, out["threshold"][t, a], thr_of(a))
            assert (r is not None) == bool(out["valid"][t, a]), (name, a, t, ct)
            if r is None:
                seen["none"] += 1
                continue
            seen["tsc" if ct == tm.TSC else "rach"] += 1
            assert r[1] == out["rssi"][t, a] and r[2] == out["timing"][t, a], (name, a, t, r[1:], out["rssi"][t, a], out["timing"][t, a])
            assert np.array_equal(np.asarray(r[0][:148], np.float32), out["soft"][t, a]), (name, a, t, ct)
    return seen


@pytest.mark.parametrize("sps,leg,frames,dense", [(1, 0, 200, None), (1, 0, 200, 0), (4, 1, 60, None)])
def test_group_equals_single_objects_and_model(pkg, sps, leg, frames, dense, request):
    # dense = 0: the equalising leg's channel estimates through the lane-per-burst kernel, the route for calls with MANY marked bursts
    # (TRXSIG_TUNE_EQ_DENSE, library-wide: the number of marked bursts above which it takes over from the wave-per-burst kernel; default 4096)
    if dense is not None:
        knob = pkg.TrxSig(sps, 0)
        knob.set_tuning(eq_dense=dense)
        request.addfinalizer(lambda: (knob.set_tuning(eq_dense=4096), knob.close()))
    S, fn0, tn0 = 128, 1000, 3
    n_slots = 8 * frames
    x, ctype = build_cells(sps, S, n_slots, fn0, tn0, seed=100 + sps)
    out, responses, final_thr = run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls=(1, 7, 8, 64, 400, 3, 123, 16, 700))   # (>= 128 slots: the replay runs parallel in time, 8 segments; >= 512: 16)
    thr = out["threshold"][~np.isnan(out["threshold"])]
    assert thr.min() < 200 and thr.max() > 255                           # the thresholds really moved both ways
    assert out["valid"].sum() > 0.2 * (ctype == tm.TSC).sum()

    # (1) S independent single-burst objects, the drop-in form of pullRadioVector
    objs = [pkg.TrxHost(sps, 0, start=(fn0, tn0), tsc_leg=leg) for _ in range(S)]
    for a in range(S):
        assert configure(objs[a].control, a) == responses[a]
    seen = check_against("single", lambda a, b, tn, fn: objs[a].pull_radio_vector(b, tn, fn), ctype, x, sps, out, range(S), fn0, tn0,
                         lambda a: objs[a].energy_threshold)
    assert seen["tsc"] > 5000 and seen["rach"] > 300 and seen["none"] > 5000, seen
    assert np.array_equal(final_thr, np.array([o.energy_threshold for o in objs]))
    for o in objs:
        o.close()

    # (2) the restatement of Transceiver.cpp on the CPU oracle
    o = oraclebind.Oracle(sps)
    models = {}
    arfcns = range(S) if sps == 1 else range(0, S, 4)
    for a in arfcns:
        models[a] = tm.TransceiverModel(o, start=(fn0, tn0), need_dfe=(leg == 0))
        assert configure(models[a].control, a) == responses[a]
    seen = check_against("model", lambda a, b, tn, fn: models[a].pull_radio_vector(b, tn, fn), ctype, x, sps, out, arfcns, fn0, tn0,
                         lambda a: models[a].energy_threshold)
    assert seen["tsc"] > 1000 and seen["rach"] > 100, seen


@pytest.mark.parametrize("sps,leg,tn0,wrap", [(4, 1, 3, False), (4, 1, 0, True), (1, 0, 6, False)])
def test_both_forms_of_the_state_machine_give_the_same(pkg, sps, leg, tn0, wrap, request):
    """TRXSIG_TUNE_GROUP_REPLAY: the wave-per-segment kernel that visits only the timeslots at which the state can move (the default, calls of
    up to 1,024 timeslots) against the kernels that step through every timeslot (round 4; the tests above hold BOTH against single objects
    and the model through the default) -- every output of every call bit for bit, calls of 1 ... 1,500 timeslots (one segment, a ragged
    last segment, sixteen segments, and past the wave form's limit), a silence of 60 frames and the hyperframe wrap inside a call."""
    S = 64
    frames = 470
    n_slots = 8 * frames
    fn0 = tm.HYPERFRAME - 100 if wrap else 4321
    x, ctype = build_cells(sps, S, n_slots, fn0, tn0, seed=4242 + sps + tn0, quiet_slots=(900, 1400))
    calls = (64, 1, 65, 128, 200, 1024, 9, 511, 1500, 63)
    knob = pkg.TrxSig(sps, 0)
    request.addfinalizer(lambda: (knob.set_tuning(group_replay=0), knob.close()))
    outs = []
    for form in (0, 1):
        knob.set_tuning(group_replay=form)
        outs.append(run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls=calls))
    knob.set_tuning(group_replay=0)
    (o0, r0, f0), (o1, r1, f1) = outs
    thr = o0["threshold"][~np.isnan(o0["threshold"])]
    assert thr.min() < 245 and thr.max() > 255
    assert o0["valid"].sum() > 1000
    for key in o0:
        assert np.array_equal(o0[key], o1[key], equal_nan=(key == "threshold")), key
    assert np.array_equal(f0, f1)


@pytest.mark.parametrize("tn0", [0, 5])
def test_group_across_the_hyperframe_wrap(pkg, tn0):
    """The frame number wraps (GSM::Time, hyperframe 2,715,648) in the middle of the run, with a 60-frame silence across it: the
    replay's frame difference to prevFalseDetectionTime is carried from slot to slot (k_group_replay_lean) and must wrap,
    restart at every false detection and trigger the quiet decrements exactly where FNDelta does -- thresholds, verdicts and
    soft bits against independent single-burst objects and the CPU model, calls that start on and off a frame boundary."""
    sps, leg, frames = 4, 1, 80
    S = 32 if tn0 == 0 else 100                                          # (100: two waves of lanes, the second one ragged)
    fn0 = tm.HYPERFRAME - 30
    n_slots = 8 * frames
    x, ctype = build_cells(sps, S, n_slots, fn0, tn0, seed=777 + tn0, quiet_slots=(80, 560))
    out, responses, final_thr = run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls=(5, 16, 1, 240, 33, 8, 530))       # (the wrap falls into the 240-slot call: segments start on both sides of it)
    thr = out["threshold"][~np.isnan(out["threshold"])]
    assert thr.min() < 245 and thr.max() > 255
    objs = [pkg.TrxHost(sps, 0, start=(fn0, tn0), tsc_leg=leg) for _ in range(S)]
    for a in range(S):
        assert configure(objs[a].control, a) == responses[a]
    seen = check_against("single", lambda a, b, tn, fn: objs[a].pull_radio_vector(b, tn, fn), ctype, x, sps, out, range(S), fn0, tn0,
                         lambda a: objs[a].energy_threshold)
    assert seen["tsc"] > 300 and seen["none"] > 1000, seen
    assert np.array_equal(final_thr, np.array([o.energy_threshold for o in objs]))
    for o in objs:
        o.close()
    o = oraclebind.Oracle(sps)
    arfcns = range(0, S, 3)
    models = {a: tm.TransceiverModel(o, start=(fn0, tn0), need_dfe=False) for a in arfcns}
    for a in arfcns:
        assert configure(models[a].control, a) == responses[a]
    check_against("model", lambda a, b, tn, fn: models[a].pull_radio_vector(b, tn, fn), ctype, x, sps, out, arfcns, fn0, tn0,
                  lambda a: models[a].energy_threshold)


def test_pipelined_mode_gives_the_same(pkg):
    """(The side-stream arrangement is no longer the default -- since the replay runs parallel in time, one stream is faster --
    and is selected here through trxsig_trxgroup_set_beside_rows.)  trxsig_trxgroup_set_pipelined: large pulls leave their replay running on the side stream while the next pull's detectors
    fill the other workspace set.  Every output of every call (collected after each pull, which joins) equals the default mode's,
    with large and small calls mixed (a small call replays on the context's stream and has to wait for the side stream first);
    and two pipelined pulls in a row with nothing in between leave the FIRST one's result intact (two workspace sets)."""
    import torch
    sps, leg, S, frames, fn0, tn0 = 4, 1, 128, 160, 5000, 6
    n_slots = 8 * frames
    x, ctype = build_cells(sps, S, n_slots, fn0, tn0, seed=4242, quiet_slots=(300, 740))
    calls = (450, 8, 470, 3, 349)                                        # ~58 rows per slot here: from ~420 slots on a call is "large"
    ref, _, thr_ref = run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls, beside_rows=24576)
    out, _, thr = run_group(pkg, sps, leg, S, n_slots, fn0, tn0, x, calls, pipelined=True, beside_rows=24576)
    for key in ref:
        assert np.array_equal(ref[key], out[key], equal_nan=True), key
    assert np.array_equal(thr_ref, thr)
    # back to back: pull A, pull B, then read A's device result (valid until the second pull after it) and B's through collect
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    g = pkg.TrxGroup(ctx, S, tsc_leg=leg, start=(fn0, tn0)); g.set_beside_rows(24576); g.set_pipelined(True)
    for a in range(S):
        configure(lambda c, a=a: g.control(a, c), a)
    cell = x.shape[2]
    dx = torch.from_numpy(x.view(np.float32).reshape(-1)).to("cuda:0")
    nA = nB = 512
    fnB, tnB = (fn0 + (tn0 + nA) // 8) % tm.HYPERFRAME, (tn0 + nA) % 8
    rA = g.pull(dx.data_ptr(), S * cell, cell, fn0, tn0, nA)
    rB = g.pull(dx.data_ptr() + 8 * nA * S * cell, S * cell, cell, fnB, tnB, nB)
    assert rA.n_rows >= 24576 and rB.n_rows >= 24576 and rA.d_row != rB.d_row        # both large: two workspace sets
    g.sync(); torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    def dev(ptr, n, dtype):                                              # the result's device arrays -> host
        a = np.empty(n, dtype)
        assert hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0       # hipMemcpyDeviceToHost
        return a
    single, _, _ = run_group(pkg, sps, leg, S, nA + nB, fn0, tn0, x[:nA + nB], (nA, nB))
    for r, lo in ((rA, 0), (rB, nA)):
        row = dev(r.d_row, r.n_slots * S, np.int32).reshape(r.n_slots, S)
        valid = dev(r.d_valid, r.n_rows, np.uint8)
        thr_rows = dev(r.d_threshold, r.n_rows, np.float64)
        got_valid = np.where(row >= 0, valid[np.maximum(row, 0)] != 0, False)
        assert np.array_equal(got_valid, single["valid"][lo:lo + r.n_slots])
        got_thr = np.where(row >= 0, thr_rows[np.maximum(row, 0)], np.nan)
        assert np.array_equal(got_thr, single["threshold"][lo:lo + r.n_slots], equal_nan=True)
    g.close(); ctx.close()


def test_reconfiguration_and_limits(pkg):
    """SETSLOT between pulls re-derives the row classes; arguments that cannot be honoured are refused."""
    import torch
    sps, S = 4, 5
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    g = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
    cell = CELL_SYM * sps
    xs, offs, lens, meta = synth.normal_batch(sps, 8 * S, 3, seed=5, sigmas=(0.0,), max_delay=0.5)
    x = np.zeros((8, S, cell), np.complex64)
    for t in range(8):
        for a in range(S):
            i = t * S + a
            n = (156 + (t % 4 == 0)) * sps
            x[t, a, :min(n, lens[i])] = xs[offs[i]:offs[i] + lens[i]][:n]
    dx = torch.from_numpy(x.view(np.float32).reshape(-1)).to("cuda:0")
    res = g.pull(dx, S * cell, cell, 10, 0, 8)
    assert res.n_rows == 0 and not g.collect()["valid"].any()          # every slot NONE: nothing reaches a correlator
    for a in range(S):
        assert g.control(a, "CMD SETTSC 3") == "RSP SETTSC 0 3"
        assert g.control(a, "CMD SETSLOT %d 1" % a) == "RSP SETSLOT 0 %d 1" % a
    res = g.pull(dx, S * cell, cell, 10, 0, 8)
    r = g.collect()
    assert res.n_rows == S
    for a in range(S):
        assert r["valid"][a, a] and r["valid"].sum() == S
        bits = (r["soft"][a, a] > 0.5).astype(np.uint8)
        assert np.array_equal(bits, meta["bits"][a * S + a])
        assert g.energy_threshold(a) == 249.0
    with pytest.raises(pkg.TrxSigError):
        g.pull(dx, 1 << 30, cell, 10, 0, 8)                             # offsets beyond 2^31 samples
    with pytest.raises(pkg.TrxSigError):
        g.pull(dx, S * cell, cell, 10, 8, 8)                            # TN out of range
    with pytest.raises(pkg.TrxSigError):
        pkg.TrxGroup(ctx, 4, tsc_leg=pkg.TSCLEG_EQUALIZE)                # the equalising leg needs sps == 1
    g.close(); ctx.close()


def make_scheduled_streams(sps, S, n_slots, fn0, tn0, seed):
    """int16 I/Q streams at 400 kS/s, one per ARFCN, whose timeslots carry what the ARFCN's schedule expects (a normal burst
    with its training sequence, an access burst with some delay), a burst of the other kind, noise or near silence."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(seed)
    m = tm.TransceiverModel.__new__(tm.TransceiverModel)
    out = []
    for a in range(S):
        tsc, slots = slot_config(a)
        m.chan_type = [slots.get(t, tm.NONE) for t in range(8)]
        nb_bits = synth.normal_bits(rng, n_slots, tsc)
        rb_bits = synth.rach_bits(rng, n_slots)
        nmod = synth.modulate(nb_bits, sps)                       # [n_slots, 157*sps]
        rmod = synth.modulate(rb_bits, sps)
        sig = []
        for t in range(n_slots):
            tn = (tn0 + t) % 8
            fn = fn0 + (tn0 + t) // 8
            n = (156 + (tn % 4 == 0)) * sps
            ct = m.expected_corr_type(tn, fn)
            kind = rng.integers(0, 10)
            amp = rng.uniform(600, 2500) * np.exp(2j * np.pi * rng.uniform())
            if kind == 0:
                v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rng.uniform(300, 1500)
            elif kind == 1:
                v = np.zeros(n, np.complex128)
            else:
                rach = (ct == tm.RACH) != (kind == 2)
                v = (rmod[t] if rach else nmod[t])[:n] * amp
                if rach:
                    d = int(rng.integers(0, 30)) * sps
                    v = np.concatenate([np.zeros(d), v[:n - d]])
            sig.append(v)
        sig = np.concatenate(sig)
        lo = resample_poly(sig, 96, 65 * sps)
        lo = lo + (rng.standard_normal(lo.size) + 1j * rng.standard_normal(lo.size)) * 10.0
        iq = np.empty((lo.size, 2), np.int16)
        iq[:, 0] = np.clip(np.round(lo.imag), -32768, 32767)      # the radio delivers Q first
        iq[:, 1] = np.clip(np.round(lo.real), -32768, 32767)
        out.append(iq)
    n = min(len(o) for o in out) // 864 * 864
    return np.stack([o[:n] for o in out]), n // 864


def test_group_on_the_fused_front_end(pkg):
    """trxsig_trxgroup_pull_rxfe (detectors of every class computing their samples from the int16 chunks) against the same
    streams through push / pop and trxsig_trxgroup_pull on the resampled bursts: every output identical."""
    import torch
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
    sps, S, fn0, tn0 = 4, 16, 200, 5
    n_slots_made = 8 * 40
    lpf = synth.design_lpf(961, 65 * sps)
    iq, nchunks = make_scheduled_streams(sps, S, n_slots_made, fn0, tn0, seed=21)
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    keys = ("valid", "soft", "rssi", "timing", "threshold")

    def clock(t):
        return (fn0 + (tn0 + t) // 8) % tm.HYPERFRAME, (tn0 + t) % 8

    # (a) fused: pushes of 1 .. 9 chunks
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    ga = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD, start=(fn0, tn0))
    fea = RxFrontEnd(ctx, S, lpf, max_chunks=9, start_tn=tn0)
    for a in range(S):
        configure(lambda c, a=a: ga.control(a, c), a)
    got_a = {k: [] for k in keys}
    c, t, sizes, k = 0, 0, (1, 3, 9, 2, 5), 0
    while c < nchunks:
        n = min(sizes[k % len(sizes)], nchunks - c); k += 1
        ns, res = ga.pull_rxfe(fea, d_iq[:, c * OUTCHUNK:(c + n) * OUTCHUNK], clock(t)[0])
        c += n
        if ns:
            assert res.n_slots == ns
            r = ga.collect()
            for key in keys:
                got_a[key].append(r[key])
            t += ns
    total_a = t
    # (b) through the resampled stream: push / pop, bursts repacked slot-major, trxsig_trxgroup_pull
    gb = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD, start=(fn0, tn0))
    feb = RxFrontEnd(ctx, S, lpf, max_chunks=4, start_tn=tn0)
    for a in range(S):
        configure(lambda c, a=a: gb.control(a, c), a)
    got_b = {k: [] for k in keys}
    cell = CELL_SYM * sps
    c = t = 0
    while c < nchunks:
        n = min(4, nchunks - c)
        feb.push_chunk(d_iq[:, c * OUTCHUNK:(c + n) * OUTCHUNK]); c += n
        popped = feb.pop_bursts()
        if popped is None:
            continue
        x, off, length, _ = popped
        nb = off.numel() // S
        xh = x.cpu().numpy().view(np.complex64).ravel(); offh = off.cpu().numpy(); lenh = length.cpu().numpy()
        cells = np.zeros((nb, S, cell), np.complex64)
        for s in range(S):
            for j in range(nb):
                i = s * nb + j
                cells[j, s, :lenh[i]] = xh[offh[i]:offh[i] + lenh[i]]
        dx = torch.from_numpy(cells.view(np.float32).reshape(-1)).cuda()
        fn, tn = clock(t)
        gb.pull(dx, S * cell, cell, fn, tn, nb)
        r = gb.collect()
        for key in keys:
            got_b[key].append(r[key])
        t += nb
    assert t == total_a and t > 250
    A = {k: np.concatenate(v) for k, v in got_a.items()}
    B = {k: np.concatenate(v) for k, v in got_b.items()}
    assert np.array_equal(A["valid"], B["valid"])
    assert np.array_equal(A["soft"], B["soft"]) and np.array_equal(A["rssi"], B["rssi"]) and np.array_equal(A["timing"], B["timing"])
    assert np.array_equal(A["threshold"], B["threshold"], equal_nan=True)
    # both kinds of burst came back, on their own slots
    m = tm.TransceiverModel.__new__(tm.TransceiverModel)
    n_tsc = n_rach = 0
    for a in range(S):
        _, slots = slot_config(a)
        m.chan_type = [slots.get(q, tm.NONE) for q in range(8)]
        for q in range(t):
            fn, tn = clock(q)
            ct = m.expected_corr_type(tn, fn)
            if A["valid"][q, a]:
                assert ct in (tm.TSC, tm.RACH)
                n_tsc += ct == tm.TSC; n_rach += ct == tm.RACH
    assert n_tsc > 300 and n_rach > 30, (n_tsc, n_rach)
    for a in range(S):
        assert ga.energy_threshold(a) == gb.energy_threshold(a)
    ga.close(); gb.close(); fea.close(); feb.close(); ctx.close()


def test_pipelined_mode_on_the_fused_front_end(pkg):
    """trxsig_trxgroup_pull_rxfe with trxsig_trxgroup_set_pipelined: 128 ARFCN streams x three pushes of 125 chunks (1,000 slots,
    ~59,000 rows each: large calls), the replay of push i overlapping the detectors of push i+1 -- every collected output and
    the final thresholds equal the default mode's on the same int16 streams (normal bursts at 400 kS/s, combination V on TN 0 of
    every 8th ARFCN so that the access-burst detector and false detections are in play, a stretch of silence)."""
    import torch
    from openbts_ttsou_amd.frontend import RxFrontEnd
    sps, S, K, tsc, pushes = 4, 128, 125, 2, 3
    dev = torch.device("cuda:0")
    nb = (K * pushes * 585 // 156 + 4 + 3) // 4 * 4
    x, off, length, meta = synth.normal_batch_torch(sps, S * nb, tsc, seed=77, device=dev, sigmas=(0.02, 0.05))
    hi = x.reshape(-1)[: S * (x.numel() // S)].reshape(S, -1)
    n_lo = K * pushes * 864
    t = torch.arange(n_lo, device=dev, dtype=torch.float64) * (65.0 * sps / 96.0)
    i0 = t.floor().long().clamp(max=hi.shape[1] - 2); fr = (t - i0).to(torch.float32)
    lo = hi[:, i0] * (1 - fr) + hi[:, i0 + 1] * fr
    lo = lo * (8000.0 / lo.abs().amax(dim=1, keepdim=True))
    lo[:, 100 * 864:170 * 864] *= 1e-3                                   # ~150 ms of near silence: the thresholds decay
    iq = torch.stack([lo.imag, lo.real], dim=2).round().clamp(-32768, 32767).to(torch.int16).contiguous()
    lpf = synth.design_lpf(961, 65 * sps)
    outs = []
    for piped, split in ((False, False), (True, False), (False, True)):
        # (third pass: trxsig_trxgroup_set_split_rows -- the access-burst class first and BESIDE the normal-burst classes, which run on the
        #  side stream, instead of in series with them)
        ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
        g = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD, start=(0, 0))
        if split:
            g.set_split_rows(16384)
        g.set_beside_rows(24576)                                         # (the side-stream arrangement, no longer the default)
        fe = RxFrontEnd(ctx, S, lpf, max_chunks=K)
        for a in range(S):
            g.control(a, "CMD SETTSC %d" % tsc)
            for tn in range(8):
                g.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 8 == 0) else 1))
        if piped:
            g.set_pipelined(True)
        got, slots = [], 0
        if piped:                                                        # two pushes in flight before anything is collected
            n0, r0 = g.pull_rxfe(fe, iq[:, :K * 864], 0)
            n1, r1 = g.pull_rxfe(fe, iq[:, K * 864:2 * K * 864], (n0 // 8) % tm.HYPERFRAME)
            assert r0.n_rows >= 24576 and r0.d_row != r1.d_row
            got.append(g.collect()); slots = n0 + n1                     # (the second push's; the first is checked through the thresholds)
            n2, _ = g.pull_rxfe(fe, iq[:, 2 * K * 864:], (slots // 8) % tm.HYPERFRAME)
            got.append(g.collect())
        else:
            for p in range(pushes):
                n, _ = g.pull_rxfe(fe, iq[:, p * K * 864:(p + 1) * K * 864], (slots // 8) % tm.HYPERFRAME)
                slots += n
                if p >= 1:
                    got.append(g.collect())
        thr = np.array([g.energy_threshold(a) for a in range(S)])
        outs.append((got, thr))
        g.close(); fe.close() if hasattr(fe, "close") else None; ctx.close()
    (ga, ta), (gb, tb), (gc, tc) = outs
    assert np.array_equal(ta, tb) and np.array_equal(ta, tc)
    for ra, rb, rc in zip(ga, gb, gc):
        assert ra["valid"].sum() > 1000
        for key in ra:
            assert np.array_equal(ra[key], rb[key], equal_nan=True), key
            assert np.array_equal(ra[key], rc[key], equal_nan=True), key


def test_group_on_a_front_end_at_one_sample_per_symbol(pkg, golden):
    """The reference's own configuration -- RadioInterface's 65 : 96 resampler with ITS filter (createLPF(., 961, 65): the
    reference table, tests/golden/resample.npz) at one sample per symbol, Transceiver with the equaliser -- through
    trxsig_trxgroup_pull_rxfe, which there goes through the resampled stream (push + pop + trxsig_trxgroup_pull_bursts on the
    listed bursts): every output equals push / pop on a second front end, the bursts repacked into cells on the host and
    trxsig_trxgroup_pull; and both normal and access bursts came back equalised / demodulated."""
    import torch
    from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
    sps, S, fn0, tn0 = 1, 24, 300, 2
    ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
    h = pkg.TrxHost(sps, 0)
    lpf = h.create_lpf(golden("resample.npz")["sendLPF_961_raw"], 65.0)        # radioInterface.cpp:230-234 at sps 1
    h.close()
    iq, nchunks = make_scheduled_streams(sps, S, 8 * 60, fn0, tn0, seed=33)
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    keys = ("valid", "soft", "rssi", "timing", "threshold")

    def clock(t):
        return (fn0 + (tn0 + t) // 8) % tm.HYPERFRAME, (tn0 + t) % 8

    def group():
        g = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_EQUALIZE, start=(fn0, tn0))
        for a in range(S):
            configure(lambda c, a=a: g.control(a, c), a)
        return g
    ga, fea = group(), RxFrontEnd(ctx, S, lpf, max_chunks=9, start_tn=tn0)
    got_a = {k: [] for k in keys}
    c, t, sizes, k = 0, 0, (1, 3, 9, 2, 5), 0
    while c < nchunks:
        n = min(sizes[k % len(sizes)], nchunks - c); k += 1
        ns, res = ga.pull_rxfe(fea, d_iq[:, c * OUTCHUNK:(c + n) * OUTCHUNK], clock(t)[0])
        c += n
        if ns:
            assert res.n_slots == ns
            r = ga.collect()
            for key in keys:
                got_a[key].append(r[key])
            t += ns
    total_a = t
    gb, feb = group(), RxFrontEnd(ctx, S, lpf, max_chunks=4, start_tn=tn0)
    got_b = {k: [] for k in keys}
    cell = CELL_SYM * sps
    c = t = 0
    while c < nchunks:
        n = min(4, nchunks - c)
        feb.push_chunk(d_iq[:, c * OUTCHUNK:(c + n) * OUTCHUNK]); c += n
        popped = feb.pop_bursts()
        if popped is None:
            continue
        x, off, length, _ = popped
        nb = off.numel() // S
        xh = x.cpu().numpy().view(np.complex64).ravel(); offh = off.cpu().numpy(); lenh = length.cpu().numpy()
        cells = np.zeros((nb, S, cell), np.complex64)
        for s in range(S):
            for j in range(nb):
                i = s * nb + j
                cells[j, s, :lenh[i]] = xh[offh[i]:offh[i] + lenh[i]]
        dx = torch.from_numpy(cells.view(np.float32).reshape(-1)).cuda()
        fn, tn = clock(t)
        gb.pull(dx, S * cell, cell, fn, tn, nb)
        r = gb.collect()
        for key in keys:
            got_b[key].append(r[key])
        t += nb
    assert t == total_a and t > 400
    A = {k: np.concatenate(v) for k, v in got_a.items()}
    B = {k: np.concatenate(v) for k, v in got_b.items()}
    for key in keys:
        assert np.array_equal(A[key], B[key], equal_nan=True), key
    assert A["valid"].sum() > 0.2 * A["valid"].size
    ga.close(); gb.close(); fea.close(); feb.close(); ctx.close()   # (the front ends before their context)
