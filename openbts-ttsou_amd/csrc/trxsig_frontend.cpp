// trxsig_frontend.cpp -- include/trxsig_frontend.h: RadioInterface's receive front end (pullBuffer + driveReceiveRadio)
// and transmit back end (driveTransmitRadio + pushBuffer) for S independent ARFCN streams, as host objects that own
// per-stream linear device buffers and enqueue one fused kernel per push / pop (k_resample, trxsig_tx.hip).  The host
// side only keeps read / write positions and the TN schedule; no sample ever visits the host.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "trxsig_ctx.h"
#include "trxsig_frontend.h"
#include "trxsig_launch.h"

namespace {
#define FE_HIP(c, call)                                                        \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return trx_ctx_fail((c), TRXSIG_EHIP, #call, e_);    \
  } while (0)

struct Guard {
  int prev = -1;
  explicit Guard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
inline int burst_len(int tn, int sps) { return (156 + ((tn & 3) == 0)) * sps; }   // radioInterface.cpp:370-378
}  // namespace

struct trxsig_rxfe {
  trxsig_ctx *c = nullptr;
  int S = 0, sps = 0, P = 0, L = 0, swap = 1, max_chunks = 0, tn = 0;
  int n_in = 0, n_out = 0, skip = 0, per_chunk = 0;         // window in, window out, INHISTORY, kept outputs per chunk
  long long stride = 0;                                     // samples per stream in the receive buffer
  int rd = 0, wr = 0;                                       // unsliced samples are [rd, wr) of every stream's row
  trx_c32 *d_rcv = nullptr, *d_tmp = nullptr;
  short2 *d_hist = nullptr;
  float *d_lpf = nullptr;
  int32_t *d_idx = nullptr;                                 // off[S*nb] then len[S*nb]
  int idx_cap = 0;
  // fused mode (trxsig_rxfe_push_detect_demod_normal): no receive buffer; what is kept between calls is raw
  int mode = 0;                                             // 0 undecided, 1 push / pop, 2 fused
  short2 *d_keep = nullptr;                                 // [S][n_in]: the window (history + chunk) of the last chunk received
  float4 *d_tpb = nullptr;                                  // [P] branch-major taps
  int tail = 0;                                             // resampled samples of earlier pushes not yet cut into bursts
  // wideband input (the channeliser, trxsig_rxfe_create_wideband): Sw raw streams at Cw x 400 kS/s, C carriers each;
  // output stream s = carrier s % C of raw stream s / C
  int Cw = 0, C = 0, Sw = 0;
  float *d_freq = nullptr;
  std::vector<float> h_freq;                                // the carrier frequencies as given (trxsig_rxfe_set_shared_filter checks their grid)
  int shared = 0;                                           // 1: the shared-filter form (trxsig_chan.hip)
  float2 *d_tw = nullptr;                                   // [C][16] exp(-j theta_c j)
  unsigned long long binmap = 0;                            // four bits per carrier: k_c, theta_c = 2 pi k_c / 16
  long long n_total = 0;                                    // raw samples (per wideband stream) received so far, offset by the history
};

// a burst whose samples (partly) lie in the send buffer: fused mode keeps its bits, not its samples
struct TxBurst { long long start; int len, slot, guard, has_gain; };

struct trxsig_txbe {
  trxsig_ctx *c = nullptr;
  int S = 0, sps = 0, Q = 0, L = 0, inchunk = 0, inhist = 0, max_bursts = 0;
  float gain = 13500.0f;
  // fused mode (the default): modulate -> resample -> gain -> int16 in ONE kernel per pop, straight from the bursts' bits; the
  // complex float32 send buffer does not exist.  The ring keeps the bits / gains of every burst that still has samples in
  // [history | pending], the host keeps where each of them starts (window coordinates: sample 0 = first history sample).
  int fused = 1, started = 0;
  int ring_cap = 0, ring_head = 0;
  uint8_t *d_ring = nullptr;                                // [S][ring_cap][148]
  float *d_rgain = nullptr;                                 // [S][ring_cap]
  int32_t *d_tab = nullptr;                                 // start[tab_cap] then meta[tab_cap]
  int tab_cap = 0;
  TrxPinRing tab_up;                                        // the table's way up (pinned: trxsig_ctx.h)
  std::vector<TxBurst> live;
  long long stride = 0, iq_stride = 0;
  int fill = 0;                                             // modulated samples behind the history, per stream
  int cur = 0;                                              // which of the two send buffers is live
  trx_c32 *d_send[2] = {nullptr, nullptr};                  // [S][inhist + capacity]: history first
  float *d_lpf = nullptr;
  short2 *d_iq = nullptr;
  int32_t *d_meta = nullptr;                                // guard[S*nb] then off[S*nb]
};

extern "C" {

int trxsig_rxfe_create(trxsig_rxfe **out, trxsig_ctx *c, int n_streams, int max_chunks, const float *h_lpf, int L, int swap_iq,
                       int start_tn) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  if (!c) return TRXSIG_EINVAL;
  if (n_streams <= 0 || n_streams > 65535 || max_chunks <= 0 || max_chunks > 65535 || !h_lpf || L <= 0 || start_tn < 0 || start_tn > 7)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_create: bad argument", hipSuccess);
  trxsig_rxfe *fe = new (std::nothrow) trxsig_rxfe;
  if (!fe) return TRXSIG_ENOMEM;
  fe->c = c; trx_ctx_retain(c); fe->S = n_streams; fe->sps = trxsig_sps(c); fe->P = 65 * fe->sps; fe->L = L; fe->swap = swap_iq != 0;
  fe->max_chunks = max_chunks; fe->tn = start_tn;
  fe->n_in = TRXSIG_OUTHISTORY + TRXSIG_OUTCHUNK;
  fe->n_out = trxsig_resample_out_len(fe->n_in, fe->P, TRXSIG_OUTRATE);
  fe->skip = 2 * fe->P;                                     // INHISTORY (radioInterface.h:38)
  fe->per_chunk = fe->n_out - fe->skip;
  fe->stride = ((long long)157 * fe->sps + (long long)max_chunks * fe->per_chunk + 63) & ~63LL;
  // burst offsets (k_burst_index) and burst counts (S * bursts per push) are 32-bit
  if (fe->stride * fe->S > 0x7fffffffLL || ((long long)max_chunks * fe->per_chunk / (156 * fe->sps) + 2) * fe->S > 0x7fffffffLL) {
    const int rc = trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_create: n_streams x max_chunks exceeds 2^31 samples (32-bit burst offsets): use several front ends", hipSuccess);
    trx_ctx_release(c);                                     // the reference taken above (error text first: the release may be the context's end)
    delete fe;
    return rc;
  }
  Guard g(trxsig_device(c));
  const size_t rcv_b = sizeof(trx_c32) * (size_t)fe->stride * fe->S, tmp_b = sizeof(trx_c32) * (size_t)157 * fe->sps * fe->S;
  if (hipMalloc((void **)&fe->d_rcv, rcv_b) != hipSuccess || hipMalloc((void **)&fe->d_tmp, tmp_b) != hipSuccess ||
      hipMalloc((void **)&fe->d_hist, sizeof(short2) * TRXSIG_OUTHISTORY * (size_t)fe->S) != hipSuccess ||
      hipMalloc((void **)&fe->d_lpf, sizeof(float) * (size_t)L) != hipSuccess ||
      hipMemset(fe->d_hist, 0, sizeof(short2) * TRXSIG_OUTHISTORY * (size_t)fe->S) != hipSuccess ||   // rcvHistory->fill(0) (:238-241)
      hipMemcpy(fe->d_lpf, h_lpf, sizeof(float) * (size_t)L, hipMemcpyHostToDevice) != hipSuccess) {
    const int rc = trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_rxfe_create: device allocation failed", hipSuccess);   // (first the error text: the release below may be the context's end)
    trxsig_rxfe_destroy(fe);
    return rc;
  }
  if (fe->sps == 4 && L <= 4 * fe->P) {                     // the fused call's tables (sps 4, at most four taps per output)
    // only the 65 branches 4 m occur (96 = 4*24, 260 = 4*65); slot n holds branch 4*(24 n mod 65) (trxsig_rxgen.h)
    std::vector<float> tpb(4 * 65, 0.0f);
    for (int n = 0; n < 65; n++) {
      const int br = 4 * ((24 * n) % 65);
      for (int k = 0; k < 4; k++) if (br + fe->P * k < L) tpb[4 * n + k] = h_lpf[br + fe->P * k];
    }
    if (hipMalloc((void **)&fe->d_tpb, sizeof(float4) * 65) != hipSuccess ||
        hipMemcpy(fe->d_tpb, tpb.data(), sizeof(float4) * 65, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&fe->d_keep, sizeof(short2) * (size_t)fe->n_in * fe->S) != hipSuccess ||
        hipMemset(fe->d_keep, 0, sizeof(short2) * (size_t)fe->n_in * fe->S) != hipSuccess) {
      const int rc = trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_rxfe_create: device allocation failed", hipSuccess);   // (first the error text: the release below may be the context's end)
      trxsig_rxfe_destroy(fe);
      return rc;
    }
  }
  *out = fe;
  return TRXSIG_OK;
}

int trxsig_rxfe_create_wideband(trxsig_rxfe **out, trxsig_ctx *c, int n_wide_streams, int n_carriers, const float *h_carrier_freq,
                                int rate_factor, int max_chunks, const float *h_lpf, int L, int swap_iq, int start_tn) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  if (!c) return TRXSIG_EINVAL;
  if (n_wide_streams <= 0 || n_carriers <= 0 || n_carriers > 64 || rate_factor <= 0 || rate_factor > 64 || !h_carrier_freq ||
      (long long)n_wide_streams * n_carriers > 65535)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_create_wideband: bad argument", hipSuccess);
  for (int k = 0; k < n_carriers; k++)
    if (!(h_carrier_freq[k] >= -3.2f && h_carrier_freq[k] <= 3.2f))
      return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_create_wideband: carrier frequencies are radians per wideband sample, |f| <= pi", hipSuccess);
  int rc = trxsig_rxfe_create(out, c, n_wide_streams * n_carriers, max_chunks, h_lpf, L, swap_iq, start_tn);
  if (rc != TRXSIG_OK) return rc;
  trxsig_rxfe *fe = *out;
  fe->Cw = rate_factor; fe->C = n_carriers; fe->Sw = n_wide_streams;
  fe->h_freq.assign(h_carrier_freq, h_carrier_freq + n_carriers);
  if (trxsig_resample_out_len(fe->n_in * rate_factor, fe->P, TRXSIG_OUTRATE * rate_factor) != fe->n_out) {
    const int rc = trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_create_wideband: this rate factor changes the resampler's output length", hipSuccess);   // (first the error text: the release below may be the context's end)
    trxsig_rxfe_destroy(fe);
    *out = nullptr;
    return rc;
  }
  fe->n_total = (long long)TRXSIG_OUTHISTORY * rate_factor;   // (the zero history in front of the stream holds raw samples 0 .. hist - 1)
  Guard g(trxsig_device(c));
  // the history is per wideband stream and rate_factor times longer; the fused (narrowband) tables are not used
  (void)hipFree(fe->d_hist); fe->d_hist = nullptr;
  (void)hipFree(fe->d_keep); fe->d_keep = nullptr; (void)hipFree(fe->d_tpb); fe->d_tpb = nullptr;
  const size_t hb = sizeof(short2) * (size_t)TRXSIG_OUTHISTORY * rate_factor * n_wide_streams;
  if (hipMalloc((void **)&fe->d_hist, hb) != hipSuccess || hipMemset(fe->d_hist, 0, hb) != hipSuccess ||
      hipMalloc((void **)&fe->d_freq, sizeof(float) * (size_t)n_carriers) != hipSuccess ||
      hipMemcpy(fe->d_freq, h_carrier_freq, sizeof(float) * (size_t)n_carriers, hipMemcpyHostToDevice) != hipSuccess) {
    const int rc = trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_rxfe_create_wideband: device allocation failed", hipSuccess);   // (first the error text: the release below may be the context's end)
    trxsig_rxfe_destroy(fe);
    *out = nullptr;
    return rc;
  }
  return TRXSIG_OK;
}

int trxsig_rxfe_push_wideband(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!fe->Cw) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_wideband: not a wideband front end", hipSuccess);
  if (!d_iq || n_chunks <= 0 || n_chunks > fe->max_chunks) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_wideband: bad argument", hipSuccess);
  fe->mode = 1;
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  const int left = fe->wr - fe->rd;
  if (fe->rd > 0 && left <= 157 * fe->sps) {                // the uncut tail moves to the front of every row (as trxsig_rxfe_push)
    if (left > 0) {
      FE_HIP(c, hipMemcpy2DAsync(fe->d_tmp, sizeof(trx_c32) * (size_t)157 * fe->sps, fe->d_rcv + fe->rd, sizeof(trx_c32) * (size_t)fe->stride,
                                 sizeof(trx_c32) * (size_t)left, fe->S, hipMemcpyDeviceToDevice, st));
      FE_HIP(c, hipMemcpy2DAsync(fe->d_rcv, sizeof(trx_c32) * (size_t)fe->stride, fe->d_tmp, sizeof(trx_c32) * (size_t)157 * fe->sps,
                                 sizeof(trx_c32) * (size_t)left, fe->S, hipMemcpyDeviceToDevice, st));
    }
    fe->rd = 0; fe->wr = left;
  }
  if ((long long)fe->wr + (long long)n_chunks * fe->per_chunk > fe->stride)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_wideband: receive buffers full (call trxsig_rxfe_pop first)", hipSuccess);
  const int Cw = fe->Cw, chunk = TRXSIG_OUTCHUNK * Cw, hist = TRXSIG_OUTHISTORY * Cw;
  TrxResampleArgs a = {};
  a.in = d_iq; a.in_stride = (long long)n_chunks * chunk; a.hist = fe->d_hist; a.hist_len = hist;
  a.n = hist + chunk; a.win_step = chunk; a.swap = fe->swap;
  a.lpf = fe->d_lpf; a.L = fe->L; a.P = fe->P; a.Q = TRXSIG_OUTRATE * Cw;
  a.o_skip = fe->skip; a.n_out = fe->n_out;
  a.out = fe->d_rcv + fe->wr; a.out_stride = fe->stride; a.out_win_step = fe->per_chunk;
  a.mix_freq = fe->d_freq; a.mix_carriers = fe->C; a.mix_n0 = fe->n_total; a.mix_tables = (const TrxTables *)trxsig_tables_device(c);
  if (fe->shared) FE_HIP(c, trx_launch_channelise16(st, a, fe->Sw, fe->C, n_chunks, fe->d_tw, trx_ctx_profiler(c), fe->binmap));
  else FE_HIP(c, trx_launch_resample_ex(st, a, fe->S, n_chunks, true, false, trx_ctx_profiler(c)));
  const short2 *tail = reinterpret_cast<const short2 *>(d_iq) + ((size_t)n_chunks * chunk - hist);
  FE_HIP(c, hipMemcpy2DAsync(fe->d_hist, sizeof(short2) * (size_t)hist, tail, sizeof(short2) * (size_t)n_chunks * chunk,
                             sizeof(short2) * (size_t)hist, fe->Sw, hipMemcpyDeviceToDevice, st));
  fe->wr += n_chunks * fe->per_chunk;
  fe->n_total += (long long)n_chunks * chunk;
  return TRXSIG_OK;
}

int trxsig_rxfe_set_shared_filter(trxsig_rxfe *fe, int on) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!fe->Cw) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_set_shared_filter: not a wideband front end", hipSuccess);
  if (!on) { fe->shared = 0; return TRXSIG_OK; }
  const int C = fe->C;
  if (!(C == 1 || C == 2 || C == 4 || C == 8 || C == 16))
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_set_shared_filter: 1, 2, 4, 8 or 16 carriers", hipSuccess);
  if ((fe->L + fe->P - 1) / fe->P > 32 || (TRXSIG_OUTCHUNK * fe->Cw) % 16 || (TRXSIG_OUTHISTORY * fe->Cw) % 16)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_set_shared_filter: needs at most 32 taps per output and chunks of a multiple of 16 samples", hipSuccess);
  // every carrier on the grid of sixteenths of the wideband rate: theta_c = 2 pi k_c / 16 (to float accuracy)
  std::vector<float2> tw((size_t)C * 16);
  unsigned long long binmap = 0;
  for (int k = 0; k < C; k++) {
    const double bins = (double)fe->h_freq[(size_t)k] / (2.0 * M_PI / 16.0);
    const double kb = std::nearbyint(bins);
    if (std::fabs(bins - kb) > 1e-5)
      return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_set_shared_filter: a carrier frequency is not a multiple of 2 pi / 16 rad per sample", hipSuccess);
    const unsigned long long bin = (unsigned long long)((((long long)kb % 16) + 16) % 16);
    for (int q = 0; q < k; q++)
      if (((binmap >> (4 * q)) & 15ull) == bin)
        return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_set_shared_filter: two carriers on one sixteenth of the wideband rate", hipSuccess);
    binmap |= bin << (4 * k);
    for (int j = 0; j < 16; j++) {
      const double ph = 2.0 * M_PI * std::fmod(kb * j, 16.0) / 16.0;
      tw[(size_t)k * 16 + j] = make_float2((float)std::cos(ph), (float)-std::sin(ph));   // exp(-j theta_c j)
    }
  }
  Guard g(trxsig_device(c));
  if (!fe->d_tw && hipMalloc((void **)&fe->d_tw, sizeof(float2) * tw.size()) != hipSuccess)
    return trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_rxfe_set_shared_filter: device allocation failed", hipSuccess);
  FE_HIP(c, hipMemcpy(fe->d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
  fe->binmap = binmap;
  fe->shared = 1;
  return TRXSIG_OK;
}

void trxsig_rxfe_destroy(trxsig_rxfe *fe) {
  if (!fe) return;
  {
    Guard g(trxsig_device(fe->c));
    (void)hipFree(fe->d_freq); (void)hipFree(fe->d_tw);
    (void)hipFree(fe->d_rcv); (void)hipFree(fe->d_tmp); (void)hipFree(fe->d_hist); (void)hipFree(fe->d_lpf); (void)hipFree(fe->d_idx);
    (void)hipFree(fe->d_keep); (void)hipFree(fe->d_tpb);
  }
  trx_ctx_release(fe->c);
  delete fe;
}

int trxsig_rxfe_pending(const trxsig_rxfe *fe) { return fe ? (fe->mode == 2 ? fe->tail : fe->wr - fe->rd) : TRXSIG_EINVAL; }

int trxsig_rxfe_push(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!d_iq || n_chunks <= 0 || n_chunks > fe->max_chunks) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push: bad argument", hipSuccess);
  if (fe->mode == 2) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push: this front end is used through the fused call", hipSuccess);
  if (fe->Cw) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push: a wideband front end takes trxsig_rxfe_push_wideband", hipSuccess);
  fe->mode = 1;
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  const int left = fe->wr - fe->rd;
  if (fe->rd > 0 && left <= 157 * fe->sps) {
    // the uncut tail (less than a burst after a pop) moves to the front of every row: two small strided copies
    if (left > 0) {
      FE_HIP(c, hipMemcpy2DAsync(fe->d_tmp, sizeof(trx_c32) * (size_t)157 * fe->sps, fe->d_rcv + fe->rd, sizeof(trx_c32) * (size_t)fe->stride,
                                 sizeof(trx_c32) * (size_t)left, fe->S, hipMemcpyDeviceToDevice, st));
      FE_HIP(c, hipMemcpy2DAsync(fe->d_rcv, sizeof(trx_c32) * (size_t)fe->stride, fe->d_tmp, sizeof(trx_c32) * (size_t)157 * fe->sps,
                                 sizeof(trx_c32) * (size_t)left, fe->S, hipMemcpyDeviceToDevice, st));
    }
    fe->rd = 0; fe->wr = left;
  }
  if ((long long)fe->wr + (long long)n_chunks * fe->per_chunk > fe->stride)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push: receive buffers full (call trxsig_rxfe_pop first)", hipSuccess);
  TrxResampleArgs a = {};
  a.in = d_iq; a.in_stride = (long long)n_chunks * TRXSIG_OUTCHUNK; a.hist = fe->d_hist; a.hist_len = TRXSIG_OUTHISTORY;
  a.n = fe->n_in; a.win_step = TRXSIG_OUTCHUNK; a.swap = fe->swap;
  a.lpf = fe->d_lpf; a.L = fe->L; a.P = fe->P; a.Q = TRXSIG_OUTRATE;
  a.o_skip = fe->skip; a.n_out = fe->n_out;
  a.out = fe->d_rcv + fe->wr; a.out_stride = fe->stride; a.out_win_step = fe->per_chunk;
  FE_HIP(c, trx_launch_resample_ex(st, a, fe->S, n_chunks, true, false, trx_ctx_profiler(c)));
  // rcvHistory = the last OUTHISTORY samples received (:259)
  const short2 *tail = reinterpret_cast<const short2 *>(d_iq) + ((size_t)n_chunks * TRXSIG_OUTCHUNK - TRXSIG_OUTHISTORY);
  FE_HIP(c, hipMemcpy2DAsync(fe->d_hist, sizeof(short2) * TRXSIG_OUTHISTORY, tail, sizeof(short2) * (size_t)n_chunks * TRXSIG_OUTCHUNK,
                             sizeof(short2) * TRXSIG_OUTHISTORY, fe->S, hipMemcpyDeviceToDevice, st));
  fe->wr += n_chunks * fe->per_chunk;
  return TRXSIG_OK;
}

int trxsig_rxfe_pop(trxsig_rxfe *fe, const trxsig_c32 **d_samples, const int32_t **d_offset, const int32_t **d_length, int32_t *h_tn,
                    int cap_tn, int *n_bursts) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!d_samples || !d_offset || !d_length || !n_bursts) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_pop: bad argument", hipSuccess);
  if (fe->mode == 2) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_pop: this front end is used through the fused call", hipSuccess);
  int nb = 0, pos = 0, tn = fe->tn;
  const int avail = fe->wr - fe->rd;
  while (avail - pos > burst_len(tn, fe->sps)) {            // "while (rcvSz > burst size)" (:375)
    if (h_tn && nb < cap_tn) h_tn[nb] = tn;
    pos += burst_len(tn, fe->sps); tn = (tn + 1) & 7; nb++;
  }
  if (h_tn && nb > cap_tn) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_pop: h_tn too small", hipSuccess);
  *n_bursts = nb;
  *d_samples = (const trxsig_c32 *)fe->d_rcv;
  if (nb == 0) { *d_offset = *d_length = nullptr; return TRXSIG_OK; }
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  if (fe->S * nb > fe->idx_cap) {
    FE_HIP(c, hipStreamSynchronize(st));
    (void)hipFree(fe->d_idx); fe->d_idx = nullptr; fe->idx_cap = 0;
    const int cap = fe->S * nb + 1024;
    FE_HIP(c, hipMalloc((void **)&fe->d_idx, sizeof(int32_t) * 2 * (size_t)cap));
    fe->idx_cap = cap;
  }
  int32_t *off = fe->d_idx, *len = fe->d_idx + fe->idx_cap;
  FE_HIP(c, trx_launch_burst_index(st, fe->S, nb, fe->stride, fe->rd, fe->tn, fe->sps, off, len));
  *d_offset = off; *d_length = len;
  fe->rd += pos; fe->tn = tn;
  return TRXSIG_OK;
}

}  // extern "C"

trxsig_ctx *trx_rxfe_ctx(trxsig_rxfe *fe) { return fe ? fe->c : nullptr; }
int trx_rxfe_streams(const trxsig_rxfe *fe) { return fe ? fe->S : 0; }
int trx_rxfe_next_tn(const trxsig_rxfe *fe) { return fe ? fe->tn : 0; }   // TN of the next burst a pop will cut

// the bursts a fused push completes and where the kernels find their samples
int trx_rxfe_fused_begin(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, TrxRxfePush *out) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!d_iq || n_chunks <= 0 || n_chunks > fe->max_chunks || !out)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "fused receive front end: bad argument", hipSuccess);
  if (!fe->d_tpb) return trx_ctx_fail(c, TRXSIG_EINVAL, "fused receive front end: needs sps == 4 and a filter of at most 4*260 taps", hipSuccess);
  if (fe->mode == 1) return trx_ctx_fail(c, TRXSIG_EINVAL, "fused receive front end: this front end is used through push / pop", hipSuccess);
  // "while (rcvSz > burst size)" (:375) over the uncut tail plus the new samples
  const int avail = fe->tail + n_chunks * fe->per_chunk;
  int nb = 0, pos = 0, tn = fe->tn;
  while (avail - pos > burst_len(tn, fe->sps)) { pos += burst_len(tn, fe->sps); tn = (tn + 1) & 7; nb++; }
  TrxRxGen gen = {};
  gen.raw = reinterpret_cast<const short2 *>(d_iq); gen.raw_stride = (long long)n_chunks * TRXSIG_OUTCHUNK;
  gen.keep = fe->d_keep; gen.tpb = fe->d_tpb; gen.K = n_chunks; gen.swap = fe->swap;
  gen.skipD = fe->skip + (fe->L - 1) / 2 / TRXSIG_OUTRATE;
  // inOff = (skipD + r)*96/260 reaches the window's end (n_in) at r = rl0 and n_in + 1 at r = rl1
  const int rl0 = (fe->n_in * fe->P + TRXSIG_OUTRATE - 1) / TRXSIG_OUTRATE - gen.skipD;
  const int rl1 = ((fe->n_in + 1) * fe->P + TRXSIG_OUTRATE - 1) / TRXSIG_OUTRATE - gen.skipD;
  gen.w0 = fe->per_chunk - rl0 > 0 ? fe->per_chunk - rl0 : 0;
  gen.w1 = fe->per_chunk - rl1 > 0 ? fe->per_chunk - rl1 : 0;
  gen.tail = fe->tail; gen.tn0 = fe->tn; gen.nb = nb; gen.sel = nullptr;
  out->gen = gen; out->nb = nb; out->tn0 = fe->tn; out->n_streams = fe->S;
  return TRXSIG_OK;
}

// once the kernels that read d_iq / the kept window are enqueued: keep the window of the last chunk, advance the clock
int trx_rxfe_fused_end(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, const TrxRxfePush &p) {
  trxsig_ctx *c = fe->c;
  fe->mode = 2;
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  // [its 192-sample history | the chunk] (stream-ordered behind the kernels that read d_keep)
  const size_t kb = sizeof(short2) * (size_t)fe->n_in, rowb = sizeof(short2) * (size_t)n_chunks * TRXSIG_OUTCHUNK;
  const short2 *raw = reinterpret_cast<const short2 *>(d_iq);
  if (n_chunks >= 2) {
    FE_HIP(c, hipMemcpy2DAsync(fe->d_keep, kb, raw + ((size_t)n_chunks * TRXSIG_OUTCHUNK - fe->n_in), rowb, kb, fe->S, hipMemcpyDeviceToDevice, st));
  } else {
    FE_HIP(c, hipMemcpy2DAsync(fe->d_keep, kb, fe->d_keep + TRXSIG_OUTCHUNK, kb, sizeof(short2) * TRXSIG_OUTHISTORY, fe->S, hipMemcpyDeviceToDevice, st));
    FE_HIP(c, hipMemcpy2DAsync(fe->d_keep + TRXSIG_OUTHISTORY, kb, raw, rowb, sizeof(short2) * TRXSIG_OUTCHUNK, fe->S, hipMemcpyDeviceToDevice, st));
  }
  int pos = 0, tn = p.tn0;
  for (int j = 0; j < p.nb; j++) { pos += burst_len(tn, fe->sps); tn = (tn + 1) & 7; }
  fe->tail = fe->tail + n_chunks * fe->per_chunk - pos;
  fe->tn = tn;
  return TRXSIG_OK;
}

extern "C" {

int trxsig_rxfe_push_detect_demod_normal(trxsig_rxfe *fe, const int16_t *d_iq, int n_chunks, int tsc, float detect_thresh,
                                         float energy_thresh, uint8_t *d_flags, trxsig_c32 *d_amp, float *d_toa, float *d_avgpwr,
                                         float *d_soft, uint8_t *d_hard, int nsoft, int soft_stride, int32_t *h_tn, int cap_tn,
                                         int *n_bursts) {
  if (!fe) return TRXSIG_EINVAL;
  trxsig_ctx *c = fe->c;
  if (!n_bursts) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_detect_demod_normal: bad argument", hipSuccess);
  TrxRxfePush p;
  int rc = trx_rxfe_fused_begin(fe, d_iq, n_chunks, &p);
  if (rc != TRXSIG_OK) return rc;
  // nothing of the front end's state has changed yet: a refused call leaves it as it was
  if (p.nb > cap_tn) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_rxfe_push_detect_demod_normal: cap_tn is smaller than the bursts this push completes (the output arrays hold n_streams * cap_tn entries)", hipSuccess);
  if (h_tn) for (int j = 0; j < p.nb; j++) h_tn[j] = (p.tn0 + j) & 7;
  *n_bursts = p.nb;
  if (p.nb > 0) {
    rc = trx_ctx_rx_normal(c, p.gen, fe->S * p.nb, tsc, detect_thresh, energy_thresh, d_flags, d_amp, d_toa, d_avgpwr, d_soft, d_hard, nsoft,
                           soft_stride);
    if (rc != TRXSIG_OK) return rc;
  }
  return trx_rxfe_fused_end(fe, d_iq, n_chunks, p);
}

// ---------------------------------------------------------------------------------------------------------------------
int trxsig_txbe_create(trxsig_txbe **out, trxsig_ctx *c, int n_streams, int max_bursts, const float *h_lpf, int L, float gain) {
  if (!out) return TRXSIG_EINVAL;
  *out = nullptr;
  if (!c) return TRXSIG_EINVAL;
  if (n_streams <= 0 || n_streams > 65535 || max_bursts <= 0 || !h_lpf || L <= 0)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_create: bad argument", hipSuccess);
  trxsig_txbe *be = new (std::nothrow) trxsig_txbe;
  if (!be) return TRXSIG_ENOMEM;
  be->c = c; trx_ctx_retain(c); be->S = n_streams; be->sps = trxsig_sps(c); be->Q = 65 * be->sps; be->L = L; be->gain = gain;
  be->inchunk = 9 * be->Q; be->inhist = 2 * be->Q; be->max_bursts = max_bursts;
  // room for one chunk of left-over plus one push of the longest bursts
  const long long cap = (long long)be->inchunk + (long long)max_bursts * 157 * be->sps;
  be->stride = (be->inhist + cap + 63) & ~63LL;
  be->iq_stride = ((long long)trxsig_resample_out_len((int)be->stride, TRXSIG_OUTRATE, be->Q) + 63) & ~63LL;
  Guard g(trxsig_device(c));
  bool ok = true;
  for (int k = 0; k < 2 && ok; k++) {
    ok = hipMalloc((void **)&be->d_send[k], sizeof(trx_c32) * (size_t)be->stride * be->S) == hipSuccess &&
         hipMemset(be->d_send[k], 0, sizeof(trx_c32) * (size_t)be->stride * be->S) == hipSuccess;   // sendHistory starts as zeros
  }
  // the ring holds what one push can add plus whatever can still be pending or in the history (less than a chunk + the history)
  be->ring_cap = max_bursts + (be->inchunk + be->inhist) / (148 * be->sps) + 4;
  be->tab_cap = be->ring_cap + 2;
  ok = ok && hipMalloc((void **)&be->d_lpf, sizeof(float) * (size_t)L) == hipSuccess &&
       hipMemcpy(be->d_lpf, h_lpf, sizeof(float) * (size_t)L, hipMemcpyHostToDevice) == hipSuccess &&
       hipMalloc((void **)&be->d_iq, sizeof(short2) * (size_t)be->iq_stride * be->S) == hipSuccess &&
       hipMalloc((void **)&be->d_meta, sizeof(int32_t) * 2 * (size_t)max_bursts * be->S) == hipSuccess &&
       hipMalloc((void **)&be->d_ring, (size_t)148 * be->ring_cap * be->S) == hipSuccess &&
       hipMalloc((void **)&be->d_rgain, sizeof(float) * (size_t)be->ring_cap * be->S) == hipSuccess &&
       hipMalloc((void **)&be->d_tab, sizeof(int32_t) * 2 * (size_t)be->tab_cap) == hipSuccess && be->ring_cap < 65536;
  if (!ok) {
    const int rc = trx_ctx_fail(c, TRXSIG_EHIP, "trxsig_txbe_create: device allocation failed", hipSuccess);   // (first the error text: the release below may be the context's end)
    trxsig_txbe_destroy(be);
    return rc;
  }
  *out = be;
  return TRXSIG_OK;
}

void trxsig_txbe_destroy(trxsig_txbe *be) {
  if (!be) return;
  {
    Guard g(trxsig_device(be->c));
    (void)hipFree(be->d_send[0]); (void)hipFree(be->d_send[1]); (void)hipFree(be->d_lpf); (void)hipFree(be->d_iq); (void)hipFree(be->d_meta);
    (void)hipFree(be->d_ring); (void)hipFree(be->d_rgain); (void)hipFree(be->d_tab);
    be->tab_up.release();
  }
  trx_ctx_release(be->c);
  delete be;
}

int trxsig_txbe_pending(const trxsig_txbe *be) { return be ? be->fill : TRXSIG_EINVAL; }

int trxsig_txbe_set_fused(trxsig_txbe *be, int fused) {
  if (!be) return TRXSIG_EINVAL;
  if (be->started) return trx_ctx_fail(be->c, TRXSIG_EINVAL, "trxsig_txbe_set_fused: the back end is in use", hipSuccess);
  be->fused = fused != 0;
  return TRXSIG_OK;
}

int trxsig_txbe_streams(const trxsig_txbe *be) { return be ? be->S : TRXSIG_EINVAL; }
extern "C++" { trxsig_ctx *trx_txbe_context(const trxsig_txbe *be) { return be ? be->c : nullptr; } }

// everything trxsig_txbe_push_bursts can refuse, checked without touching the back end (the Transceiver group asks BEFORE it pops
// its transmit queue: a refused push must not have consumed the bursts)
int trxsig_txbe_can_push(trxsig_txbe *be, const int32_t *h_guard, int n_bursts) {
  if (!be) return TRXSIG_EINVAL;
  trxsig_ctx *c = be->c;
  if (!h_guard || n_bursts <= 0 || n_bursts > be->max_bursts)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_push_bursts: bad argument", hipSuccess);
  long long tot = 0;
  for (int j = 0; j < n_bursts; j++) {
    if (h_guard[j] < 0 || h_guard[j] > 9) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_push_bursts: guard must be 0..9", hipSuccess);
    tot += (long long)be->sps * (148 + h_guard[j]);
  }
  if (be->inhist + be->fill + tot > be->stride)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_push_bursts: send buffers full (call trxsig_txbe_pop first)", hipSuccess);
  if (be->fused && (int)be->live.size() + n_bursts > be->ring_cap)
    return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_push_bursts: burst ring full (call trxsig_txbe_pop first)", hipSuccess);
  return TRXSIG_OK;
}

int trxsig_txbe_push_bursts(trxsig_txbe *be, const uint8_t *d_bits, const int32_t *h_guard, const float *d_gain, int n_bursts) {
  if (!be) return TRXSIG_EINVAL;
  trxsig_ctx *c = be->c;
  if (!d_bits) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_push_bursts: bad argument", hipSuccess);
  const int ok = trxsig_txbe_can_push(be, h_guard, n_bursts);
  if (ok != TRXSIG_OK) return ok;
  long long tot = 0;
  for (int j = 0; j < n_bursts; j++) tot += (long long)be->sps * (148 + h_guard[j]);
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  be->started = 1;
  if (be->fused) {
    // only the bits travel: burst j of this push takes ring slot head + j and starts where the pending samples end
    FE_HIP(c, trx_launch_tx_ring_store(st, d_bits, d_gain, be->S, n_bursts, be->ring_head, be->ring_cap, be->d_ring, be->d_rgain));
    long long pos = (long long)be->inhist + be->fill;
    for (int j = 0; j < n_bursts; j++) {
      TxBurst b;
      b.start = pos; b.len = be->sps * (148 + h_guard[j]); b.slot = (be->ring_head + j) % be->ring_cap; b.guard = h_guard[j];
      b.has_gain = d_gain != nullptr;
      be->live.push_back(b);
      pos += b.len;
    }
    be->ring_head = (be->ring_head + n_bursts) % be->ring_cap;
    be->fill += (int)tot;
    return TRXSIG_OK;
  }
  const int B = be->S * n_bursts;
  std::vector<int32_t> meta(2 * (size_t)B);
  for (int s = 0; s < be->S; s++) {
    long long pos = (long long)s * be->stride + be->inhist + be->fill;
    for (int j = 0; j < n_bursts; j++) {
      meta[(size_t)s * n_bursts + j] = h_guard[j];
      meta[(size_t)B + (size_t)s * n_bursts + j] = (int32_t)pos;
      pos += (long long)be->sps * (148 + h_guard[j]);
    }
  }
  FE_HIP(c, hipMemcpyAsync(be->d_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice, st));
  FE_HIP(c, hipStreamSynchronize(st));                      // meta is a local
  int rc = trxsig_modulate_batch(c, d_bits, be->d_meta, d_gain, B, (trxsig_c32 *)be->d_send[be->cur], be->d_meta + B);
  if (rc != TRXSIG_OK) return rc;
  be->fill += (int)tot;
  return TRXSIG_OK;
}

int trxsig_txbe_pop(trxsig_txbe *be, const int16_t **d_iq, int64_t *stream_stride, int *n_samples) {
  if (!be) return TRXSIG_EINVAL;
  trxsig_ctx *c = be->c;
  if (!d_iq || !stream_stride || !n_samples) return trx_ctx_fail(c, TRXSIG_EINVAL, "trxsig_txbe_pop: bad argument", hipSuccess);
  *d_iq = (const int16_t *)be->d_iq; *stream_stride = be->iq_stride; *n_samples = 0;
  const int nch = be->fill / be->inchunk;                   // "if (sendBuffer->size() < INCHUNK) return" (:125-127)
  if (nch == 0) return TRXSIG_OK;
  const int ntr = nch * be->inchunk;                        // truncatedBuffer (:131-132)
  const int n_in = be->inhist + ntr;                        // signalVector(*sendHistory, *truncatedBuffer) (:141)
  const int n_out = trxsig_resample_out_len(n_in, TRXSIG_OUTRATE, be->Q);
  Guard g(trxsig_device(c));
  hipStream_t st = (hipStream_t)trxsig_get_stream(c);
  TrxResampleArgs a = {};
  a.in = be->d_send[be->cur]; a.in_stride = be->stride; a.n = n_in;
  a.lpf = be->d_lpf; a.L = be->L; a.P = TRXSIG_OUTRATE; a.Q = be->Q;
  a.o_skip = TRXSIG_OUTHISTORY; a.n_out = n_out;           // writeSamples(resampledVectorShort + OUTHISTORY*2, size - OUTHISTORY) (:165-166)
  a.out = be->d_iq; a.out_stride = be->iq_stride; a.gain = be->gain;
  if (be->fused) {
    // the window [history | whole chunks] as a list of bursts: where each starts, its ring slot, guard and gain flag
    const int M = (int)be->live.size();
    void *tab_p = nullptr;
    int tab_slot = 0;
    FE_HIP(c, be->tab_up.take(sizeof(int32_t) * 2 * (size_t)be->tab_cap, &tab_p, &tab_slot));
    int32_t *tab = (int32_t *)tab_p;
    std::memset(tab, 0, sizeof(int32_t) * 2 * (size_t)be->tab_cap);
    int m_used = 0;
    for (int m = 0; m < M && m_used < be->tab_cap; m++) {
      const TxBurst &b = be->live[(size_t)m];
      if (b.start >= n_in) break;                           // lies entirely in the left-over
      tab[(size_t)m_used] = (int32_t)b.start;
      tab[(size_t)be->tab_cap + m_used] = b.slot | (b.guard << 16) | (b.has_gain << 20);
      m_used++;
    }
    FE_HIP(c, be->tab_up.upload(tab_slot, be->d_tab, sizeof(int32_t) * 2 * (size_t)be->tab_cap, st));
    a.in = be->d_ring; a.in_stride = be->ring_cap;
    a.tx_tables = (const TrxTables *)trxsig_tables_device(c); a.tx_gain = be->d_rgain; a.tx_start = be->d_tab; a.tx_meta = be->d_tab + be->tab_cap;
    a.tx_n = m_used; a.tx_sps = be->sps;
    FE_HIP(c, trx_launch_resample_ex(st, a, be->S, 1, false, true, trx_ctx_profiler(c), true));
    // sendHistory = the last INHISTORY samples sent, the rest of sendBuffer follows it (:183-191): the window's origin moves on
    // by ntr samples; bursts that end before it are done
    size_t keep_from = 0;
    for (size_t m = 0; m < be->live.size(); m++) {
      be->live[m].start -= ntr;
      if (be->live[m].start + be->live[m].len <= 0) keep_from = m + 1;
    }
    be->live.erase(be->live.begin(), be->live.begin() + (long)keep_from);
    be->fill -= ntr;
    *n_samples = n_out - TRXSIG_OUTHISTORY;
    return TRXSIG_OK;
  }
  FE_HIP(c, trx_launch_resample_ex(st, a, be->S, 1, false, true, trx_ctx_profiler(c)));
  // sendHistory = the last INHISTORY samples sent (:183-184), the rest of sendBuffer follows it (:187-191): into the other buffer
  const int keep = be->inhist + be->fill - ntr;             // history + left-over, contiguous at [ntr, ntr + keep)
  FE_HIP(c, hipMemcpy2DAsync(be->d_send[be->cur ^ 1], sizeof(trx_c32) * (size_t)be->stride, be->d_send[be->cur] + ntr,
                             sizeof(trx_c32) * (size_t)be->stride, sizeof(trx_c32) * (size_t)keep, be->S, hipMemcpyDeviceToDevice, st));
  be->cur ^= 1;
  be->fill -= ntr;
  *n_samples = n_out - TRXSIG_OUTHISTORY;
  return TRXSIG_OK;
}

}  // extern "C"
