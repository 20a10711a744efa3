"""Synthetic complex-baseband bursts (test and benchmark INPUT generation only).

These generators produce the workloads of SURVEY 8d (configs 2 and 3): 148-bit GSM bursts, GMSK
modulated at `sps` samples/symbol, with random complex gain, sub-sample delay and AWGN, packed as
628/624/624/624-sample bursts (157-156-156-156 symbols, Transceiver/radioInterface.cpp:370-378).
The modulator here is a plain numpy/torch model of GMSK (ideal i^k rotation); it does not need to
match the reference's modulateBurst bit for bit because it only manufactures *inputs* -- the same
arrays are fed to the HIP path and to the oracle.
"""
import numpy as np

TRAINING_SEQUENCE = [  # GSM 05.02 training sequence codes (bit values as in GSM/GSMCommon.cpp:44-53)
    "00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110",
    "01000111101101000100011110", "00011010111001000001101011", "01001110101100000100111010",
    "10100111110110001010011111", "11101111000100101110111100"]
RACH_SYNCH = "01001011011111111001100110101010001111000"     # GSM/GSMCommon.cpp:57


def gsm_pulse(sps):
    """The reference's GSM pulse approximation 0.96*exp(-1.1380 t^2 - 0.527 t^4), 2*sps+1 taps,
    normalised to unit energy per symbol (Transceiver/sigProcLib.cpp:411-430)."""
    t = (np.arange(2 * sps + 1) - sps) / float(sps)
    p = 0.96 * np.exp(-1.1380 * t ** 2 - 0.527 * t ** 4)
    return (p / np.sqrt((p ** 2).sum() / sps)).astype(np.float32)


def burst_lengths(B, sps):
    """628/624/624/624-style lengths: guard 9 symbols when index % 4 == 0, else 8."""
    guard = np.where(np.arange(B) % 4 == 0, 9, 8)
    length = ((148 + guard) * sps).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int64)
    return guard, length, off


def normal_bits(rng, B, tsc):
    bits = rng.integers(0, 2, (B, 148)).astype(np.uint8)
    bits[:, :3] = 0
    bits[:, -3:] = 0
    bits[:, 61:87] = np.array([int(c) for c in TRAINING_SEQUENCE[tsc]], np.uint8)
    return bits


def rach_bits(rng, B):
    bits = np.zeros((B, 148), np.uint8)
    bits[:, :8] = [0, 1, 0, 1, 0, 1, 0, 1]
    bits[:, 8:49] = np.array([int(c) for c in RACH_SYNCH], np.uint8)
    bits[:, 49:85] = rng.integers(0, 2, (B, 36))
    return bits


def modulate(bits, sps, nsym=157):
    """bits [B,148] -> complex64 [B, nsym*sps] GMSK baseband (unit amplitude)."""
    B = bits.shape[0]
    k = np.arange(148)
    sym = (2.0 * bits - 1.0) * (1j ** (k % 4))[None, :]
    up = np.zeros((B, nsym * sps + 2 * sps), np.complex64)
    up[:, sps:sps + 148 * sps:sps] = sym
    p = gsm_pulse(sps)
    out = np.zeros((B, nsym * sps), np.complex64)
    for j in range(2 * sps + 1):                      # out[t] = sum_j p[j] a[t + sps - j]
        out += p[j] * up[:, 2 * sps - j:2 * sps - j + nsym * sps]
    return out


def _delay(x, d):
    """Per-row delay by d samples (d may be fractional / large): integer shift + 21-tap sinc."""
    B, N = x.shape
    di = np.floor(d).astype(np.int64)
    fr = (d - di).astype(np.float64)
    j = np.arange(21)
    taps = np.sinc(j[None, :] - 10 - fr[:, None])        # sinc(j - 10 - frac)
    pad = np.zeros((B, N + 20), np.complex64)
    pad[:, 10:10 + N] = x
    y = np.zeros((B, N), np.complex64)
    for jj in range(21):                                 # y[t] = sum_j taps[j] x[t + 10 - j]
        y += (taps[:, jj:jj + 1] * pad[:, 20 - jj:20 - jj + N]).astype(np.complex64)
    out = np.zeros_like(y)
    for s in np.unique(di):
        rows = np.flatnonzero(di == s)
        if s >= 0:
            if s < N: out[rows, s:] = y[rows, :N - s]
        else:
            if -s < N: out[rows, :N + s] = y[rows, -s:]
    return out


def _finish(rng, base, B, sps, delay, sigma_choices):
    guard, length, off = burst_lengths(B, sps)
    amp = (rng.uniform(300, 3000, B) * np.exp(2j * np.pi * rng.uniform(size=B))).astype(np.complex64)
    sig = np.asarray(sigma_choices, np.float32)[np.arange(B) % len(sigma_choices)]
    x = np.zeros(int(length.sum()), np.complex64)
    CH = 512
    for s in range(0, B, CH):
        e = min(B, s + CH)
        y = _delay(base[s:e], delay[s:e]) * amp[s:e, None]
        n = (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape)) / np.sqrt(2.0)
        y = (y + (sig[s:e] * np.abs(amp[s:e]))[:, None] * n).astype(np.complex64)
        for i in range(s, e):
            x[off[i]:off[i] + length[i]] = y[i - s, :length[i]]
    return x, off.astype(np.int32), length, amp, sig


def bursts_from_bits(bits, sps, seed=0, sigmas=(0.0, 0.1, 0.3), max_delay=1.5):
    """Caller-chosen 148-bit bursts [B,148] through the same channel model as normal_batch."""
    rng = np.random.default_rng(seed)
    B = bits.shape[0]
    delay = rng.uniform(-max_delay, max_delay, B)
    x, off, length, amp, sig = _finish(rng, modulate(bits, sps), B, sps, delay, sigmas)
    return x, off, length, dict(bits=bits, amp=amp, delay=delay.astype(np.float32), sigma=sig)


def normal_batch(sps, B, tsc, seed=0, sigmas=(0.0, 0.1, 0.3), max_delay=1.5):
    """Config 2 workload: B normal bursts with training sequence `tsc`."""
    rng = np.random.default_rng(seed)
    bits = normal_bits(rng, B, tsc)
    delay = rng.uniform(-max_delay, max_delay, B)
    x, off, length, amp, sig = _finish(rng, modulate(bits, sps), B, sps, delay, sigmas)
    return x, off, length, dict(bits=bits, amp=amp, delay=delay.astype(np.float32), sigma=sig)


def rach_batch(sps, B, seed=0, sigmas=(0.0, 0.1, 0.3), max_delay_sym=60):
    """Config 3 workload: B access bursts arriving 0..max_delay_sym symbols (+ fraction) late."""
    rng = np.random.default_rng(seed)
    bits = rach_bits(rng, B)
    delay = rng.integers(0, max_delay_sym + 1, B) * sps + rng.uniform(size=B)
    x, off, length, amp, sig = _finish(rng, modulate(bits, sps), B, sps, delay, sigmas)
    return x, off, length, dict(bits=bits, amp=amp, delay=delay.astype(np.float32), sigma=sig)


# ---- torch (device) generators for the full-size benchmark workloads -----------------------------
def _torch_batch(bits, sps, delay, amp, sigma, device, gen):
    """bits [B,148] uint8 (torch, device) -> packed complex64 bursts on `device` (628/624/624/624)."""
    import torch
    B = bits.shape[0]
    nsym = 157
    N = nsym * sps
    k = torch.arange(148, device=device)
    rot = torch.tensor([1, 1j, -1, -1j], dtype=torch.complex64, device=device)[k % 4]
    sym = (2.0 * bits.to(torch.float32) - 1.0).to(torch.complex64) * rot[None, :]
    up = torch.zeros(B, N + 2 * sps, dtype=torch.complex64, device=device)
    up[:, sps:sps + 148 * sps:sps] = sym
    p = torch.from_numpy(gsm_pulse(sps)).to(device)
    base = torch.zeros(B, N, dtype=torch.complex64, device=device)
    for j in range(2 * sps + 1):
        base += p[j] * up[:, 2 * sps - j:2 * sps - j + N]
    # delay: integer part by gather, fractional part by 21-tap sinc
    di = torch.floor(delay)
    fr = (delay - di).to(torch.float32)
    j = torch.arange(21, device=device, dtype=torch.float32)
    taps = torch.sinc(j[None, :] - 10 - fr[:, None])
    pad = torch.zeros(B, N + 20, dtype=torch.complex64, device=device)
    pad[:, 10:10 + N] = base
    y = torch.zeros(B, N, dtype=torch.complex64, device=device)
    for jj in range(21):
        y += taps[:, jj:jj + 1] * pad[:, 20 - jj:20 - jj + N]
    idx = torch.arange(N, device=device)[None, :] - di.to(torch.int64)[:, None]
    valid = (idx >= 0) & (idx < N)
    y = torch.where(valid, torch.gather(y, 1, idx.clamp(0, N - 1)), torch.zeros((), dtype=torch.complex64, device=device))
    noise = torch.randn(B, N, 2, device=device, generator=gen)
    noise = torch.view_as_complex(noise) * (0.70710678 * sigma * amp.abs())[:, None]
    y = y * amp[:, None] + noise
    guard, length, off = burst_lengths(B, sps)
    keep = torch.arange(N, device=device)[None, :] < torch.from_numpy(length).to(device)[:, None]
    x = y[keep].contiguous()                         # row-major: burst after burst
    return x, torch.from_numpy(off.astype(np.int32)).to(device), torch.from_numpy(length).to(device)


def normal_batch_torch(sps, B, tsc, seed=0, device="cuda:0", sigmas=(0.0, 0.1, 0.3), max_delay=1.5,
                       chunk=8192):
    """Config 2 workload generated on the GPU: returns (x complex64 packed, off int32, len int32, meta)
    with meta = dict(bits uint8 [B,148], amp complex64 [B], delay float32 [B], sigma float32 [B])."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    tsb = torch.tensor([int(c) for c in TRAINING_SEQUENCE[tsc]], dtype=torch.uint8, device=dev)
    xs, bits_all, amp_all, delay_all, sig_all = [], [], [], [], []
    sig_choices = torch.tensor(sigmas, dtype=torch.float32, device=dev)
    assert chunk % 4 == 0
    for s in range(0, B, chunk):
        n = min(chunk, B - s)
        bits = torch.randint(0, 2, (n, 148), device=dev, generator=gen, dtype=torch.uint8)
        bits[:, :3] = 0; bits[:, -3:] = 0; bits[:, 61:87] = tsb
        mag = 300 + 2700 * torch.rand(n, device=dev, generator=gen)
        ph = 2 * np.pi * torch.rand(n, device=dev, generator=gen)
        amp = torch.polar(mag, ph)
        delay = (2 * torch.rand(n, device=dev, generator=gen) - 1) * max_delay
        sigma = sig_choices[(torch.arange(n, device=dev) + s) % len(sigmas)]
        x, _, _ = _torch_batch(bits, sps, delay, amp, sigma, dev, gen)
        xs.append(x); bits_all.append(bits); amp_all.append(amp); delay_all.append(delay); sig_all.append(sigma)
    guard, length, off = burst_lengths(B, sps)
    x = torch.cat(xs)
    meta = dict(bits=torch.cat(bits_all), amp=torch.cat(amp_all), delay=torch.cat(delay_all),
                sigma=torch.cat(sig_all))
    return (x, torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(length).to(dev), meta)


def rach_batch_torch(sps, B, seed=0, device="cuda:0", sigmas=(0.0, 0.1, 0.3), max_delay_sym=60, chunk=8192):
    """Config 3 workload generated on the GPU (access bursts arriving 0..max_delay_sym symbols late)."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    sync = torch.tensor([int(c) for c in RACH_SYNCH], dtype=torch.uint8, device=dev)
    xs, bits_all, amp_all, delay_all, sig_all = [], [], [], [], []
    sig_choices = torch.tensor(sigmas, dtype=torch.float32, device=dev)
    for s in range(0, B, chunk):
        n = min(chunk, B - s)
        bits = torch.zeros(n, 148, dtype=torch.uint8, device=dev)
        bits[:, 1:8:2] = 1
        bits[:, 8:49] = sync
        bits[:, 49:85] = torch.randint(0, 2, (n, 36), device=dev, generator=gen, dtype=torch.uint8)
        mag = 300 + 2700 * torch.rand(n, device=dev, generator=gen)
        ph = 2 * np.pi * torch.rand(n, device=dev, generator=gen)
        amp = torch.polar(mag, ph)
        delay = (torch.randint(0, max_delay_sym + 1, (n,), device=dev, generator=gen) * sps).to(torch.float32) + \
            torch.rand(n, device=dev, generator=gen)
        sigma = sig_choices[(torch.arange(n, device=dev) + s) % len(sigmas)]
        x, _, _ = _torch_batch(bits, sps, delay, amp, sigma, dev, gen)
        xs.append(x); bits_all.append(bits); amp_all.append(amp); delay_all.append(delay); sig_all.append(sigma)
    guard, length, off = burst_lengths(B, sps)
    meta = dict(bits=torch.cat(bits_all), amp=torch.cat(amp_all), delay=torch.cat(delay_all), sigma=torch.cat(sig_all))
    return (torch.cat(xs), torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(length).to(dev), meta)


def design_lpf(L, P, beta=5.0, cutoff=0.9):
    """An L-tap Kaiser-windowed sinc low-pass for interpolation by P (cutoff at `cutoff` x the input Nyquist), DC gain P:
    what createLPF(cutoff, L, P) is meant to be.  The reference's createLPF ignores its cutoff and loads a fixed table
    designed for 65:96 (one sample per symbol, sigProcLib.cpp:1106-1139); used with P = 65*4 that table does not
    interpolate (measured with the reference's own code: the resampled signal correlates 0.58 with the original and 4 of
    29 clean bursts are detected), so the config-4 BENCHMARK feeds the resampler this filter (taps are an argument of
    the library).  The parity tests keep the reference's tables."""
    n = np.arange(L, dtype=np.float64) - (L - 1) / 2.0
    h = np.sinc(cutoff / P * n) * np.kaiser(L, beta)
    return (h * (P / h.sum())).astype(np.float32)
