"""Where does k_eq_dfe4's time go, role by role?  Needs the probe build (clock64() stamps round every workgroup barrier; the
stamps come back through the soft bits of each workgroup's first four bursts, so this build's outputs are NOT results):
    make -C openbts-ttsou_amd/csrc probe_dfe4
    TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_d4probe.so python tools/dfe4_probe.py        (-> profiles/r05_eq_probe.txt)
Per role (wave 0: the decision-feedback recursion, wave 1: the feed-forward FIR + the soft bits' way out, waves 2, 3: delayVector)
it prints the cycles a wave spends BETWEEN barriers (its own work, issue stalls and memory waits included) and AT them (waiting
for the slowest role of the step), averaged over the workgroups of config 5's 65,536-burst call."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
import bench_config5

ap = argparse.ArgumentParser(); ap.add_argument("--bursts", type=int, default=65536); a = ap.parse_args()
dev = torch.device("cuda:0")
args = argparse.Namespace(bursts=a.bursts, eq_tail=1)
wl = bench_config5.Config5(args)
ctx = pkg.TrxSig(1, 0); ctx.use_torch_stream()
wl.setup(pkg, ctx, dev, 0, args)
for _ in range(200):
    wl.step()
torch.cuda.synchronize()
s = wl.soft.cpu().numpy()
B = a.bursts
rows = s[: B // 64 * 64].reshape(-1, 64, s.shape[1])[:, :4, :4]        # [workgroup][row b0 + role][work, wait, total, role]
names = {0: "wave 0: decision-feedback recursion", 1: "wave 1: feed-forward FIR + soft bits out", 2: "wave 2: delayVector (even tiles)",
         3: "wave 3: delayVector (odd tiles, tile -1)"}
print("k_eq_dfe4, %d bursts, %d workgroups; cycles per wave (clock64), mean over the workgroups [p10 .. p90]" % (B, rows.shape[0]))
for role in range(4):
    r = rows[:, role, :]
    assert np.all(r[:, 3] == role)
    w, q, t = r[:, 0], r[:, 1], r[:, 2]
    print("%-44s work %7.0f [%6.0f .. %6.0f]   at barriers %7.0f [%6.0f .. %6.0f]   total %7.0f" % (
        names[role], w.mean(), np.percentile(w, 10), np.percentile(w, 90), q.mean(), np.percentile(q, 10), np.percentile(q, 90), t.mean()))
tot = rows[:, :, 2].mean()
print("a workgroup's wave lives %.0f cycles; 26 steps -> %.0f cycles per step; the step's length is its slowest role's work" % (tot, tot / 26.0))
