#!/usr/bin/env python3
"""Timing experiment for k_normal_chain: the detect role alone, the demodulate role alone (on tags left by the
detect-only run) and both, per lag.  Usage: python tools/chain_roles.py [lib.so ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(lib):
    if lib:                                                  # another build of the TUNING library
        import _pkg as _p
        _p.load().TUNE_LIB_PATH = lib
    import torch, _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import synth
    dev = torch.device("cuda:0")
    sps, B, tsc = 4, 65536, 2
    x, off, length, meta = synth.normal_batch_torch(sps, B, tsc, seed=0xB5E55ED0, device=dev)
    xf = torch.view_as_real(x).contiguous()
    fl = torch.zeros(B, dtype=torch.uint8, device=dev); amp = torch.zeros(B, 2, device=dev); toa = torch.zeros(B, device=dev)
    soft = torch.zeros(B, 148, device=dev)
    out = {}
    for lag in (int(a) for a in os.environ.get("LAGS", "48").split(",")):
        for name, dbg in (("detect only", 2), ("demod only", 1), ("detect only again", 2)):
            t = getattr(main, "ctx", None)
            if t is None:
                t = main.ctx = pkg.TrxSig(sps, 0, tuning=True); t.use_torch_stream(); t.set_tuning(normal_path=5)
            t.set_tuning(chain_lag=lag)
            t._chk(t.L.trxsig_set_tuning(t.h, 6, dbg), "dbg")
            def step():
                t.detect_demod_normal(xf, off, length, tsc, fl, amp, toa, soft, detect_thresh=3.0, energy_thresh=0.0, nsoft=148, soft_stride=148)
            for _ in range(300): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(500): step()
            torch.cuda.synchronize(); out["lag %d %s" % (lag, name)] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
    print(lib or "default", json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
