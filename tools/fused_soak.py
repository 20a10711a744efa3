"""Soak test of the fused receive front end (trxsig_rxfe_push_detect_demod_normal) against push + pop + detect: random
stream counts, start TNs and push sizes over a long random int16 stream (noise + bursts), every output compared bit for
bit after every push.   python tools/fused_soak.py [pushes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from openbts_ttsou_amd import synth
from openbts_ttsou_amd.frontend import RxFrontEnd, OUTCHUNK
dev = torch.device('cuda:0')
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(123)
ctx = pkg.TrxSig(4, 0); ctx.use_torch_stream()
total = 0
for trial in range(6):
    S = int(rng.integers(1, 9)); tn0 = int(rng.integers(0, 8)); tsc = int(rng.integers(0, 8)); maxk = int(rng.choice([1, 2, 5, 16]))
    lpf = synth.design_lpf(961, 260) if trial % 2 == 0 else (synth.design_lpf(961, 260, beta=3.0, cutoff=0.7))
    # content: modulated bursts (detectable) + noise, brought to 400 kS/s by linear interpolation, random gain per stream
    nchunks = NP * (maxk + 1) // 2 + maxk
    nb0 = (nchunks * 585 // 156 + 8) // 4 * 4
    x, off, length, meta = synth.normal_batch_torch(4, S * nb0, tsc, seed=1000 + trial, device=dev, sigmas=(0.02, 0.2, 1.0))
    hi = x.reshape(-1)[: S * (x.numel() // S)].reshape(S, -1)
    tt = torch.arange(nchunks * 864, device=dev, dtype=torch.float64) * (260.0 / 96.0)
    i0 = tt.floor().long().clamp(max=hi.shape[1] - 2); fr = (tt - i0).to(torch.float32)
    lo = hi[:, i0] * (1 - fr) + hi[:, i0 + 1] * fr
    lo = lo * (torch.tensor(rng.uniform(200, 20000, S), device=dev, dtype=torch.float32) / lo.abs().amax(dim=1))[:, None]
    iq = torch.stack([lo.imag, lo.real], dim=2).round().clamp(-32768, 32767).to(torch.int16).contiguous()
    fa = RxFrontEnd(ctx, S, lpf, max_chunks=maxk, start_tn=tn0)
    fb = RxFrontEnd(ctx, S, lpf, max_chunks=maxk, start_tn=tn0)
    c = 0
    for p in range(NP):
        k = int(rng.integers(1, maxk + 1))
        if (c + k) * 864 > iq.shape[1]:
            break
        seg = iq[:, c * 864:(c + k) * 864].contiguous(); c += k
        n = S * (2 + 4 * k)
        o = [dict(flags=torch.zeros(n, dtype=torch.uint8, device=dev), amp=torch.zeros(n, 2, device=dev), toa=torch.zeros(n, device=dev),
                  pwr=torch.zeros(n, device=dev), soft=torch.full((n, 148), -1.0, device=dev)) for _ in range(2)]
        fa.push_chunk(seg)
        r = fa.pop_raw()
        nb, tn = fb.push_detect_demod(seg, tsc, o[1]["flags"], o[1]["amp"], o[1]["toa"], o[1]["soft"], avgpwr=o[1]["pwr"], nsoft=148, soft_stride=148)
        assert (r is None) == (nb == 0), (trial, p)
        if r is None:
            continue
        ps, po, pl, tna, nba = r
        assert nba == nb and np.array_equal(tna, tn), (trial, p)
        ctx._chk(ctx.L.trxsig_detect_demod_normal_batch(ctx.h, ps, po, pl, S * nb, tsc, 3.0, 0.0, o[0]["flags"].data_ptr(), o[0]["amp"].data_ptr(),
                                                        o[0]["toa"].data_ptr(), o[0]["pwr"].data_ptr(), o[0]["soft"].data_ptr(), None, 148, 148), "dd")
        torch.cuda.synchronize()
        for kk in ("flags", "amp", "toa", "pwr", "soft"):
            a, b = o[0][kk][:S * nb], o[1][kk][:S * nb]
            a = a.view(torch.int32) if a.dtype == torch.float32 else a
            b = b.view(torch.int32) if b.dtype == torch.float32 else b
            assert torch.equal(a, b), (trial, p, kk, S, k, tn0)
        total += S * nb
    det = float(((o[1]["flags"][:S * nb] & pkg.F_DETECT) != 0).float().mean())
    print("trial %d: S=%d start_tn=%d tsc=%d max_chunks=%d, %d pushes identical (last push detected %.2f)" % (trial, S, tn0, tsc, maxk, p + 1, det), flush=True)
print("fused soak: %d bursts, every output bit-identical to push + pop + detect" % total)
