"""Receive front-end orchestration for S independent ARFCN streams (BASELINE config 4): the data
movement of RadioInterface::pullBuffer + driveReceiveRadio (Transceiver/radioInterface.cpp:197-273,
359-401) around libtrxsig's kernels.

Per chunk of OUTCHUNK = 864 int16 I/Q samples at 400 kS/s and per stream:
    unUSRPifyVector (int16 -> float, I/Q swapped)          -> trxsig_unpack_int16
    [192-sample history | chunk] -> polyphaseResampleVector(P = 65*sps, Q = 96, LPF)
                                                            -> trxsig_resample_batch
    drop the first INHISTORY = 130*sps outputs, append to the stream's receive buffer
then the buffer is cut into bursts of (156 + (TN % 4 == 0)) * sps samples (157-156-156-156).
Everything numeric runs in the library; this class only concatenates and slices device buffers
(torch is plumbing).  State per stream = the 192-sample history and the unsliced tail, as in the
reference.
"""
import numpy as np

OUTRATE = 96
OUTCHUNK = 9 * OUTRATE          # 864
OUTHISTORY = 2 * OUTRATE        # 192


class RxFrontEnd:
    def __init__(self, ctx, n_streams, lpf_taps, device="cuda:0", swap_iq=True):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.S = n_streams
        self.sps = ctx.sps
        self.P = 65 * self.sps
        self.inhistory = 2 * self.P
        self.dev = torch.device(device)
        self.swap = swap_iq
        self.lpf = torch.as_tensor(np.ascontiguousarray(lpf_taps, np.float32)).to(self.dev)
        self.n_in = OUTHISTORY + OUTCHUNK
        self.n_out = ctx.resample_out_len(self.n_in, self.P, OUTRATE)
        # [S, history + chunk] complex (float pairs): the history lives in the first 192 entries
        self.inbuf = torch.zeros(self.S, self.n_in, 2, dtype=torch.float32, device=self.dev)
        self.outbuf = torch.zeros(self.S, self.n_out, 2, dtype=torch.float32, device=self.dev)
        self.rcv = torch.zeros(self.S, 0, 2, dtype=torch.float32, device=self.dev)   # unsliced tail per stream
        self.tn = 0                                                                   # TN of the next burst

    def push_chunk(self, iq):
        """iq: int16 tensor [S, 864, 2] (device).  Resamples and appends to the receive buffers."""
        torch = self.torch
        assert iq.shape == (self.S, OUTCHUNK, 2) and iq.dtype == torch.int16
        chunk = torch.empty(self.S, OUTCHUNK, 2, dtype=torch.float32, device=self.dev)
        self.ctx.unpack_int16(iq.contiguous(), self.S * OUTCHUNK, chunk, swap_iq=self.swap)
        self.inbuf[:, OUTHISTORY:] = chunk
        self.ctx.resample(self.inbuf, self.n_in, self.n_in, self.S, self.P, OUTRATE, self.lpf, self.outbuf, self.n_out)
        self.rcv = torch.cat([self.rcv, self.outbuf[:, self.inhistory:]], dim=1)
        self.inbuf[:, :OUTHISTORY] = chunk[:, OUTCHUNK - OUTHISTORY:].clone()          # history for the next chunk

    def pop_bursts(self):
        """Cut every stream's buffer into bursts (same schedule on all streams).  Returns
        (samples [float pairs, packed], offset int32 [S*nb], length int32 [S*nb], tn int32 [S*nb]) with
        bursts ordered stream-major, or None when less than one burst is buffered."""
        torch = self.torch
        lens, tns = [], []
        pos, tn, avail = 0, self.tn, self.rcv.shape[1]
        while True:
            n = (156 + (tn % 4 == 0)) * self.sps
            if not (avail - pos > n):                       # "while (rcvSz > burstSize)" (:375)
                break
            lens.append(n); tns.append(tn)
            pos += n; tn = (tn + 1) % 8
        if not lens:
            return None
        used = self.rcv[:, :pos].contiguous()               # [S, pos, 2]
        self.rcv = self.rcv[:, pos:].contiguous()
        self.tn = tn
        lens = np.array(lens, np.int32)
        off1 = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        off = (np.arange(self.S)[:, None] * pos + off1[None, :]).astype(np.int32).ravel()
        length = np.tile(lens, self.S)
        tnv = np.tile(np.array(tns, np.int32), self.S)
        return (used.view(-1, 2), torch.from_numpy(off).to(self.dev), torch.from_numpy(length).to(self.dev), tnv)


INCHUNK_SYM = 65 * 9            # per sps: INCHUNK = INRATE*9 with INRATE = 65*sps (radioInterface.h)


class TxBackEnd:
    """Transmit back-end for S independent ARFCN streams: the data movement of RadioInterface::pushBuffer
    (Transceiver/radioInterface.cpp:123-194) around libtrxsig's kernels -- modulated bursts are appended
    to the send buffer; whenever it holds at least INCHUNK = 585*sps samples, [INHISTORY history | whole
    chunks] goes through polyphaseResampleVector(P = 96, Q = 65*sps, sendLPF), scaleVector(gain) and
    USRPifyVector, and the first OUTHISTORY outputs are dropped.  Numerics in the library; this class
    only concatenates and slices device buffers."""

    def __init__(self, ctx, n_streams, lpf_taps, gain=13500.0, device="cuda:0"):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.S = n_streams
        self.sps = ctx.sps
        self.Q = 65 * self.sps
        self.inchunk = INCHUNK_SYM * self.sps
        self.inhistory = 2 * self.Q
        self.gain = float(gain)
        self.dev = torch.device(device)
        self.lpf = torch.as_tensor(np.ascontiguousarray(lpf_taps, np.float32)).to(self.dev)
        self.hist = torch.zeros(self.S, self.inhistory, 2, dtype=torch.float32, device=self.dev)
        self.send = torch.zeros(self.S, 0, 2, dtype=torch.float32, device=self.dev)

    def push_bursts(self, bits, guard, gain=None):
        """bits: uint8 [S, nb, 148]; guard: int32 [nb] guard symbols per burst (same schedule on every stream);
        gain: optional float32 [S, nb] (addRadioVector's power scaling).  modulateBurst for all of them."""
        torch = self.torch
        S, nb = bits.shape[0], bits.shape[1]
        assert S == self.S
        guard = np.ascontiguousarray(guard, np.int32)
        lens = (self.sps * (148 + guard)).astype(np.int64)
        tot = int(lens.sum())
        off1 = np.concatenate([[0], np.cumsum(lens)[:-1]])
        off = (np.arange(S)[:, None] * tot + off1[None, :]).astype(np.int32).ravel()
        out = torch.zeros(S, tot, 2, dtype=torch.float32, device=self.dev)
        d_bits = torch.as_tensor(np.ascontiguousarray(bits, np.uint8).reshape(S * nb, 148)).to(self.dev)
        d_guard = torch.from_numpy(np.tile(guard, S)).to(self.dev)
        d_gain = None if gain is None else torch.as_tensor(np.ascontiguousarray(gain, np.float32).ravel()).to(self.dev)
        self.ctx.modulate(d_bits, d_guard, out, torch.from_numpy(off).to(self.dev), gain=d_gain)
        self.send = torch.cat([self.send, out], dim=1)

    def pop_samples(self):
        """int16 tensor [S, n, 2] for the radio (96/(65*sps) samples per modulator sample), or None while
        less than one chunk is buffered."""
        torch = self.torch
        nch = self.send.shape[1] // self.inchunk
        if nch == 0:
            return None
        ntr = nch * self.inchunk
        inp = torch.cat([self.hist, self.send[:, :ntr]], dim=1).contiguous()        # [S, INHISTORY + ntr, 2]
        n_in = inp.shape[1]
        n_out = self.ctx.resample_out_len(n_in, OUTRATE, self.Q)
        res = torch.zeros(self.S, n_out, 2, dtype=torch.float32, device=self.dev)
        self.ctx.resample(inp, n_in, n_in, self.S, OUTRATE, self.Q, self.lpf, res, n_out)
        iq = torch.zeros(self.S, n_out, 2, dtype=torch.int16, device=self.dev)
        self.ctx.pack_int16_scaled(res, self.S * n_out, self.gain, iq)
        self.hist = self.send[:, ntr - self.inhistory:ntr].clone()
        self.send = self.send[:, ntr:].contiguous()
        return iq[:, OUTHISTORY:]
