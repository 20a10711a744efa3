"""Side measurement, the TRANSMIT half of the Transceiver group: per step the GSM core's datagrams for F frames of S ARFCNs
(every timeslot carries a burst) are added (trxsig_trxgroup_add_bursts: host parse + sort, one upload, queue insertion on the
device) and the same F frames are pushed straight into the fused transmit back end (trxsig_trxgroup_push_txbe: queue / stale
dump / filler table on the device, then ONE kernel bits -> modulate -> resample -> int16 at the pop).
    python tools/group_tx_bench.py [S] [frames per step] [staged|copy]
(staged, the default since round 5: the datagrams are written into the group's pinned staging block -- where a host would
 recvfrom() them; copy: handed over in a pageable array, which the call copies first)
us_per_step is the loop's wall time per step, everything included (the ~30 us of numpy that write the frame numbers too); the
host_us_* fields split the HOST's time by call -- trxsig_trxgroup_tx_staging is where the host is held back when the device is
more than a batch behind; us_per_step_synchronised is a step with a device synchronise behind it (its latency on an idle device)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd.frontend import TxBackEnd
from openbts_ttsou_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
STAGED = (sys.argv[3] if len(sys.argv) > 3 else "staged") == "staged"     # "copy": trxsig_trxgroup_add_bursts from a pageable array
sps = 4
lpf = synth.design_lpf(651, 96)
ctx = pkg.TrxSig(sps, 0); ctx.use_torch_stream()
grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
for a in range(S):
    for tn in range(8):
        grp.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 8 == 0) else 1))
be = TxBackEnd(ctx, S, lpf, max_bursts=8 * F)
rng = np.random.default_rng(3)
n = S * 8 * F
base = np.zeros((n, 154), np.uint8)
base[:, 6:] = rng.integers(0, 2, (n, 148))
base[:, 5] = rng.integers(0, 30, n)
arf = np.repeat(np.arange(S, dtype=np.int32), 8 * F)
tn = np.tile(np.tile(np.arange(8), F), S)
fo = np.tile(np.repeat(np.arange(F), 8), S)
base[:, 0] = tn
perm = rng.permutation(n)                                   # arrival order: ARFCNs interleaved
base, arf, fo = base[perm], arf[perm], fo[perm]
fn = 1000
t_add = t_push = t_recv = t_stage = t_pushonly = 0.0
seen = set()


K = 100
# the frame numbers of every step's datagrams, made BEFORE the timed loop (this script's own numpy work is not the library's step)
hdrs = []
for k in range(K + 10):
    f = (fn + F * k + fo).astype(np.uint32)
    hdrs.append(np.stack([f >> 24, (f >> 16) & 255, (f >> 8) & 255, f & 255], axis=1).astype(np.uint8))
step_no = 0


def step():
    global fn, t_add, t_push, t_recv, step_no
    h = hdrs[step_no]; step_no += 1
    global t_stage, t_pushonly
    ts = time.perf_counter()
    if STAGED:                                              # (the call may wait for the upload that last used this block: the library's time, counted)
        d, a = grp.tx_staging(n)
    tr = time.perf_counter()
    t_stage += tr - ts
    if STAGED:                                              # the datagrams "arrive" in the group's pinned block (a host would recvfrom() there)
        key = d.ctypes.data
        if key not in seen:                                 # (two blocks alternate: the payloads are the same every step, written once per block)
            d[:] = base; a[:] = arf; seen.add(key)
        d[:, 1:5] = h                                       # this step's frame numbers
    else:
        base[:, 1:5] = h
    t0 = time.perf_counter()
    t_recv += t0 - tr
    if STAGED:
        grp.add_staged(n)
    else:
        grp.add_bursts(base, arf)
    t1 = time.perf_counter()
    grp.push_txbe(be, fn, 0, 8 * F)
    t2 = time.perf_counter()
    iq = be.pop_samples()
    t_add += t1 - t0; t_push += time.perf_counter() - t1; t_pushonly += t2 - t1
    fn += F
    return iq

for _ in range(10): iq = step()
torch.cuda.synchronize()
t_add = t_push = t_recv = t_stage = t_pushonly = 0.0
t0 = time.perf_counter()
for _ in range(K): iq = step()
torch.cuda.synchronize()
dt_all = (time.perf_counter() - t0) / K
dt = dt_all                                                 # the step's wall time, the emulated arrival of the datagrams included (~35 us of numpy at
                                                            # 8,192 datagrams).  (Until r05 session 17 this line subtracted the time spent in the arrival
                                                            # block, which then also held trxsig_trxgroup_tx_staging's wait for the DEVICE: the 85 us steps
                                                            # that accounting reported were ~350 us steps.  rocprofv3's kernel times had said so.)
q, dropped = grp.tx_queue_size(0)
# one step at a time (a synchronize after each): the step's latency on an idle device, for comparison with the pipelined rate
K2 = 10
hdrs += hdrs[:K2]
for k in range(K2):
    f = (fn + F * k + fo).astype(np.uint32)
    hdrs[step_no + k] = np.stack([f >> 24, (f >> 16) & 255, (f >> 8) & 255, f & 255], axis=1).astype(np.uint8)
tl = 0.0
for _ in range(K2):
    ta = time.perf_counter(); step(); torch.cuda.synchronize(); tl += time.perf_counter() - ta
print(json.dumps({"arfcns": S, "frames_per_step": F, "add": "trxsig_trxgroup_add_staged (received into the pinned block)" if STAGED else "trxsig_trxgroup_add_bursts (copy from a pageable array)", "bursts_per_step": n, "us_per_step": round(dt * 1e6, 1), "host_us_emulating_arrival": round(t_recv / K * 1e6, 1),
                  "Mbursts_per_s": round(n / dt / 1e6, 2), "host_us_in_add_bursts": round(t_add / K * 1e6, 1),
                  "host_us_in_push_and_pop": round(t_push / K * 1e6, 1), "host_us_in_tx_staging": round(t_stage / K * 1e6, 1), "host_us_in_push_txbe": round(t_pushonly / K * 1e6, 1), "int16_pairs_out_per_stream": int(iq.shape[1]),
                  "us_per_step_synchronised": round(tl / K2 * 1e6, 1), "queue_left": q, "dropped": dropped,
                  "one_burst_per_call_object": "trxsig_trx_add_radio_vector + _push_radio_vector: ~40 + ~10 us per burst (tools/host_path_bench.py)"}))
