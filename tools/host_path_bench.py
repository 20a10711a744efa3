"""PCIe-inclusive rates of the host-buffer entry points (never the headline): numpy in, numpy out through
trxsig_detect_demod_normal_host -- one burst per call (the drop-in form of pullRadioVector) and 65,536 bursts per call.
    python tools/host_path_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
dev = torch.device('cuda:0')
B, sps, tsc = 65536, 4, 2
x, off, length, meta = synth.normal_batch_torch(sps, B, tsc, seed=3, device=dev)
xh = x.cpu().numpy(); offh = off.cpu().numpy(); lenh = length.cpu().numpy()
t = pkg.TrxSig(sps, 0)
out = {}
one = xh[:lenh[0]].copy(); o1 = np.zeros(1, np.int32); l1 = lenh[:1].copy()
for _ in range(200): t.detect_demod_host(one, o1, l1, tsc=tsc)
t0 = time.perf_counter()
K = 2000
for _ in range(K): t.detect_demod_host(one, o1, l1, tsc=tsc)
dt = (time.perf_counter() - t0) / K
out['one_burst_per_call'] = {'us_per_call': round(dt * 1e6, 1), 'bursts_per_s': round(1 / dt), 'gsm_slot_us': 577}
for _ in range(3): t.detect_demod_host(xh, offh, lenh, tsc=tsc)
t0 = time.perf_counter()
K = 10
for _ in range(K): r = t.detect_demod_host(xh, offh, lenh, tsc=tsc)
dt = (time.perf_counter() - t0) / K
out['65536_bursts_per_call'] = {'ms_per_call': round(dt * 1e3, 2), 'Mbursts_per_s': round(B / dt / 1e6, 2),
                                'host_to_device_MB': round(xh.nbytes / 1e6), 'device_to_host_MB': round((r['soft'].nbytes + 17 * B) / 1e6),
                                'note': 'pageable numpy buffers, includes the result allocation'}
print(json.dumps(out))

# the Transceiver object (include/trxsig_transceiver.h), one burst per call: TSC leg with the DFE cache, sps = 1
h = pkg.TrxHost(1, 0, start=(100, 0))
for c in ("CMD POWEROFF", "CMD RXTUNE 890000", "CMD TXTUNE 935000", "CMD SETTSC 6", "CMD SETRXGAIN 10", "CMD SETPOWER 0", "CMD POWERON",
          "CMD SETSLOT 1 1", "CMD SETSLOT 0 5"):
    h.control(c)
xs, offs, lens, _ = synth.normal_batch(1, 64, 6, seed=3, sigmas=(0.02,), max_delay=1.0)
bursts = [xs[offs[i]:offs[i] + lens[i]].copy() * np.float32(30) for i in range(64)]
fn = 200
for i in range(200):
    fn += 1; h.pull_radio_vector(bursts[i % 64], 1, fn)
t0 = time.perf_counter()
K, got = 2000, 0
for i in range(K):
    fn += 1
    got += h.pull_radio_vector(bursts[i % 64], 1, fn) is not None
dt = (time.perf_counter() - t0) / K
print(json.dumps({'trx_pull_radio_vector_tsc_leg': {'us_per_call': round(dt * 1e6, 1), 'soft_vectors_returned': got, 'of': K}}))
