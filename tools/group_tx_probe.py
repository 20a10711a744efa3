"""Where does the time of k_group_tx_ingest / k_group_tx_push go?  Needs the probe build (clock64() stamps at the phase boundaries,
workgroup 0's first thread):
    make -C openbts-ttsou_amd/csrc probe_tx
    TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so python tools/group_tx_probe.py [S] [frames per step]
The stamps are clock64() = s_memtime, the shader clock (~2.15 GHz under this load: k_group_tx_ingest's 999 hundreds are the 45 us
rocprofv3 gives it); the phases are printed in HUNDREDS of shader clocks (21.5 of them a microsecond)."""
import ctypes as C, os, sys
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
ctx = pkg.TrxSig(4, 0); ctx.use_torch_stream()
grp = pkg.TrxGroup(ctx, S, tsc_leg=pkg.TSCLEG_DEMOD)
for a in range(S):
    for tn in range(8):
        grp.control(a, "CMD SETSLOT %d %d" % (tn, 5 if (tn == 0 and a % 8 == 0) else 1))
be = TxBackEnd(ctx, S, synth.design_lpf(651, 96), max_bursts=8 * F)
rng = np.random.default_rng(3)
n = S * 8 * F
base = np.zeros((n, 154), np.uint8)
base[:, 6:] = rng.integers(0, 2, (n, 148)); base[:, 5] = rng.integers(0, 30, n)
arf = np.repeat(np.arange(S, dtype=np.int32), 8 * F)
fo = np.tile(np.repeat(np.arange(F), 8), S)
base[:, 0] = np.tile(np.tile(np.arange(8), F), S)
perm = rng.permutation(n)
base, arf, fo = base[perm], arf[perm], fo[perm]
L = ctx.L
L.trx_txprobe_read.argtypes = [C.c_void_p]
acc = np.zeros((2, 8)); K = 20
fn = 1000
for it in range(K + 3):
    f = (fn + fo).astype(np.uint32)
    base[:, 1:5] = np.stack([f >> 24, (f >> 16) & 255, (f >> 8) & 255, f & 255], axis=1)
    grp.add_bursts(base, arf); grp.push_txbe(be, fn, 0, 8 * F); be.pop_samples(); fn += F
    torch.cuda.synchronize()
    st = np.zeros((2, 8), np.uint64)
    assert L.trx_txprobe_read(st.ctypes.data) == 0
    if it >= 3: acc += st.astype(np.float64)
st = acc / K
def us(k, i, j): return (st[k, j] - st[k, i]) / 100.0
print("S = %d, %d frames per step (%d bursts); workgroup 0, hundreds of shader clocks (clock64; ~21.5 a microsecond), mean of %d steps" % (S, F, n, K))
def f(k1, i1, k2, i2): return (st[k2, i2] - st[k1, i1]) / 100.0
print("k_group_tx<ingest, walk> (one launch: the add call's ingest taken into the push): queues + filler tables in, the arrival kernel's totals %.1f | its lists in + free slots %.1f | "
      "the pushes (a wave per ARFCN) %.1f | wait for the slowest wave's pushes + payload stores %.1f | the walk (a wave per ARFCN; the gather its tail) %.1f, of it inside the pops %.1f | "
      "queues + filler tables back %.1f | total %.1f" % (
    f(1, 0, 0, 2), f(0, 2, 0, 3), f(0, 3, 0, 4), f(0, 4, 0, 5), f(1, 1, 1, 2), st[1, 4] / 100.0, f(1, 2, 1, 3), f(1, 0, 1, 3)))
