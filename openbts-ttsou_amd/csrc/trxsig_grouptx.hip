// trxsig_grouptx.hip -- the TRANSMIT half of the Transceiver group (include/trxsig_trxgroup.h): addRadioVector /
// pushRadioVector (Transceiver/Transceiver.cpp:100-113, 138-181) for S ARFCNs with the priority queue, the stale-burst dump
// and the filler table [FN % modulus][TN] on the device.
//   k_group_tx_add    : a lane per ARFCN takes its new bursts in arrival order: a payload slot from the ARFCN's free stack,
//                       the (time, slot) entry into its queue (trxsig_txq.h: std::priority_queue's moves);
//   k_group_tx_store  : the bursts' 148 bits + gain into their payload slots (every thread a word);
//   k_group_tx_push   : a lane per ARFCN walks n_slots timeslots: stale entries leave the queue for the filler table, the
//                       entry for exactly this time (if any) replaces the filler entry and goes out, else the filler entry
//                       goes out (:142-177) -- as payload REFERENCES, nothing is copied on the serial path;
//   k_group_tx_gather : the referenced payloads into the layout trxsig_txbe_push_bursts takes ([S][n][148] bits, [S][n]
//                       gains): what the fused transmit back end then modulates, resamples and packs to int16.
// What is kept per burst is its bits and its gain, never its modulated samples: modulateBurst + scaleVector of the same bits
// and gain give the same samples every time they are formed, so the filler table's "copy of the burst" (:165) is a reference.
#include "trxsig_dev.h"
#include "trxsig_group.h"
#include "trxsig_txq.h"

namespace {

__global__ __launch_bounds__(64) void k_group_tx_add(TrxGroupTx x, const int32_t *__restrict__ seg, const int32_t *__restrict__ s_fn,
                                                     const int32_t *__restrict__ s_tn, int32_t *__restrict__ s_pid) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= x.S) return;
  const TrxqView q = {x.q_fn + a, x.q_key + a, x.S};
  int nq = x.q_n[a], nf = x.free_n[a];
  const int j1 = seg[a + 1];
  for (int j = seg[a]; j < j1; j++) {
    if (nq >= x.qcap || nf == 0) {                          // queue or payload pool full: the burst is dropped and the ARFCN marked
      x.status[a] |= 1u;
      s_pid[j] = -1;
      continue;
    }
    nf--;
    const int pid = x.free_stack[(size_t)nf * x.S + a];
    nq = trxq_push(q, nq, s_fn[j], s_tn[j] | (pid << 3));   // mTransmitPriorityQueue.write(newVec) (:109)
    s_pid[j] = pid;
  }
  x.q_n[a] = nq;
  x.free_n[a] = nf;
}

// stage: [n][TRXG_PAYLOAD_WORDS] words (148 bits one per byte, then the gain), ARFCN-sorted like s_pid / s_arfcn
__global__ __launch_bounds__(256) void k_group_tx_store(TrxGroupTx x, int n, const int32_t *__restrict__ s_arfcn, const int32_t *__restrict__ s_pid,
                                                        const uint32_t *__restrict__ stage) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)n * TRXG_PAYLOAD_WORDS) return;
  const int j = (int)(g / TRXG_PAYLOAD_WORDS), w = (int)(g - (long long)j * TRXG_PAYLOAD_WORDS);
  const int pid = s_pid[j];
  if (pid < 0) return;
  x.pool[((size_t)s_arfcn[j] * x.npool + pid) * TRXG_PAYLOAD_WORDS + w] = stage[g];
}

__device__ __forceinline__ void tx_free(const TrxGroupTx &x, int a, int &nf, int pid) {
  if (pid < 0) return;                                      // the dummy burst is nobody's
  x.free_stack[(size_t)nf * x.S + a] = (int16_t)pid;
  nf++;
}

__global__ __launch_bounds__(64) void k_group_tx_push(TrxGroupTx x, int fn0, int tn0, int n_slots, int16_t *__restrict__ out_pid,
                                                      uint8_t *__restrict__ out_fq) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= x.S) return;
  const TrxqView q = {x.q_fn + a, x.q_key + a, x.S};
  int nq = x.q_n[a], nf = x.free_n[a];
  int mod[8];
#pragma unroll
  for (int k = 0; k < 8; k++) mod[k] = x.fmod[k * x.S + a];
  for (int t = 0; t < n_slots; t++) {
    const int tn = (tn0 + t) & 7;
    int fn = fn0 + ((tn0 + t) >> 3);
    fn -= fn >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;      // (n_slots < 8 * gHyperframe: the host checks)
    fn -= fn >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
    // dump stale bursts, if any: "even if the burst is stale, put it in the filler table" (:142-153)
    while (nq > 0 && trxq_time_lt(q.fn[0], q.key[0] & 7, fn, tn)) {
      int32_t efn, ekey;
      nq = trxq_pop(q, nq, &efn, &ekey);
      const int etn = ekey & 7;
      int16_t *cell = x.filler + ((size_t)(efn % mod[etn]) * 8 + etn) * x.S + a;
      tx_free(x, a, nf, *cell);
      *cell = (int16_t)(ekey >> 3);
    }
    int16_t *cell = x.filler + ((size_t)(fn % mod[tn]) * 8 + tn) * x.S + a;
    int fq = 0;
    if (nq > 0 && q.fn[0] == fn && (q.key[0] & 7) == tn) {  // data at the desired timestamp (:159-173)
      int32_t efn, ekey;
      nq = trxq_pop(q, nq, &efn, &ekey);
      tx_free(x, a, nf, *cell);
      *cell = (int16_t)(ekey >> 3);
      fq = 1;
    }
    out_pid[(size_t)t * x.S + a] = *cell;                   // otherwise the filler entry (:175-177)
    out_fq[(size_t)t * x.S + a] = (uint8_t)fq;
  }
  x.q_n[a] = nq;
  x.free_n[a] = nf;
}

// bits_out [S][n_slots][148], gain_out [S][n_slots], fq_out [S][n_slots] (the transposes of out_pid / out_fq's [n_slots][S])
__global__ __launch_bounds__(256) void k_group_tx_gather(TrxGroupTx x, int n_slots, const int16_t *__restrict__ out_pid,
                                                         const uint8_t *__restrict__ out_fq, uint32_t *__restrict__ bits_out,
                                                         float *__restrict__ gain_out, uint8_t *__restrict__ fq_out) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)x.S * n_slots * TRXG_PAYLOAD_WORDS) return;
  const long long cell = g / TRXG_PAYLOAD_WORDS;
  const int w = (int)(g - cell * TRXG_PAYLOAD_WORDS);
  const int a = (int)(cell / n_slots), t = (int)(cell - (long long)a * n_slots);
  const int pid = out_pid[(size_t)t * x.S + a];
  const uint32_t *src = pid < 0 ? x.dummy : x.pool + ((size_t)a * x.npool + pid) * TRXG_PAYLOAD_WORDS;
  const uint32_t v = src[w];
  if (w < 37) bits_out[cell * 37 + w] = v;
  else {
    gain_out[cell] = __uint_as_float(v);
    fq_out[cell] = out_fq[(size_t)t * x.S + a];
  }
}

}  // namespace

hipError_t trx_launch_group_tx_add(hipStream_t st, const TrxGroupTx &x, int n, const int32_t *seg, const int32_t *s_fn, const int32_t *s_tn,
                                   const int32_t *s_arfcn, int32_t *s_pid, const uint32_t *stage) {
  if (n <= 0) return hipSuccess;
  k_group_tx_add<<<dim3((x.S + 63) / 64), dim3(64), 0, st>>>(x, seg, s_fn, s_tn, s_pid);
  const long long words = (long long)n * TRXG_PAYLOAD_WORDS;
  k_group_tx_store<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(x, n, s_arfcn, s_pid, stage);
  return hipGetLastError();
}

hipError_t trx_launch_group_tx_push(hipStream_t st, const TrxGroupTx &x, int fn0, int tn0, int n_slots, int16_t *out_pid, uint8_t *out_fq,
                                    uint8_t *bits_out, float *gain_out, uint8_t *fq_out) {
  if (n_slots <= 0) return hipSuccess;
  k_group_tx_push<<<dim3((x.S + 63) / 64), dim3(64), 0, st>>>(x, fn0, tn0, n_slots, out_pid, out_fq);
  const long long words = (long long)x.S * n_slots * TRXG_PAYLOAD_WORDS;
  k_group_tx_gather<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(x, n_slots, out_pid, out_fq, (uint32_t *)bits_out, gain_out,
                                                                                   fq_out);
  return hipGetLastError();
}
