// trxsig_grouptx.hip -- the TRANSMIT half of the Transceiver group (include/trxsig_trxgroup.h): addRadioVector /
// pushRadioVector (Transceiver/Transceiver.cpp:100-113, 138-181) for S ARFCNs with the priority queue, the stale-burst dump
// and the filler table [FN % modulus][TN] on the device.
//   k_group_tx_ingest : driveTransmitPriorityQueue's parsing (:596-620) + addRadioVector (:100-113) ON THE DEVICE (round 5) from the
//                       raw 154-byte datagrams and their ARFCN ids, as they arrived: a workgroup owns sixteen ARFCNs, finds
//                       its datagrams (a stable counting sort by wave ballots: arrival order is kept inside an ARFCN),
//                       parses TN / big-endian FN / RSSI, and a lane per ARFCN enters them in its queue -- the queue
//                       (trxsig_txq.h: std::priority_queue's moves) sits in LDS for the duration, the payload slots to hand
//                       out are fetched ahead -- then every thread copies payload words (148 bits + gain) to the slots;
//   k_group_tx_push   : a lane per ARFCN walks n_slots timeslots: stale entries leave the queue for the filler table, the
//                       entry for exactly this time (if any) replaces the filler entry and goes out, else the filler entry
//                       goes out (:142-177) -- as payload REFERENCES, nothing is copied on the serial path; the sixteen
//                       ARFCNs' queues AND filler tables are in LDS for the walk (round 5: a dependent global access per
//                       queue move and per slot was the whole kernel);
//   k_group_tx_gather : the referenced payloads into the layout trxsig_txbe_push_bursts takes ([S][n][148] bits, [S][n]
//                       gains): what the fused transmit back end then modulates, resamples and packs to int16.
// What is kept per burst is its bits and its gain, never its modulated samples: modulateBurst + scaleVector of the same bits
// and gain give the same samples every time they are formed, so the filler table's "copy of the burst" (:165) is a reference.
#include "trxsig_dev.h"
#include "trxsig_group.h"
#include "trxsig_txq.h"
#include "trxsig_txq_lds.h"

#ifdef TRX_TX_PROBE
// probe build (make probe_tx, tools/group_tx_probe.py): clock64() at the phase boundaries of the two serial kernels, workgroup 0's wave 0
__device__ unsigned long long g_txprobe[2][8];
#define TX_STAMP(kern, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_txprobe[kern][i] = clock64(); } while (0)
extern "C" int trx_txprobe_read(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_txprobe), sizeof(g_txprobe)); }
#else
#define TX_STAMP(kern, i) do { } while (0)
#endif

namespace {

constexpr int kTxA = 16;                                   // ARFCNs per workgroup
constexpr int kTxQ = TRXQ_LDS_CAP;                          // queue entries per ARFCN held in LDS (= the queue's capacity, trxsig_trxgroup.cpp)
constexpr int kTxWin = 8192;                                // datagrams per round: 128 chunks of a wave's width (120 KB of LDS in all)
constexpr int kTxChunks = kTxWin / 64;

struct TxGainTab { float v[26]; };                          // pow(10, q), q = -12 .. 13 (host: the reference's double pow, rounded to float)

constexpr int kTxRow = kTxQ + 1;                            // an ARFCN's row in LDS: one entry of padding (the rows start on different banks;
                                                            // speculative reads past the queue's end land on it)

// the sixteen queues of a workgroup, LDS <-> memory (element i of ARFCN a lives at [i * S + a]; in LDS an entry is the pair (fn, key))
__device__ __forceinline__ void tx_queues_load(const TrxGroupTx &x, int a0, TrxqEnt (*q)[kTxRow], const int *nq) {
  const int k = threadIdx.x & (kTxA - 1);
  if (a0 + k < x.S)
    for (int i = threadIdx.x / kTxA; i < nq[k]; i += blockDim.x / kTxA)
      q[k][i] = trxq_ent(x.q_fn[(size_t)i * x.S + a0 + k], x.q_key[(size_t)i * x.S + a0 + k]);
}
__device__ __forceinline__ void tx_queues_store(const TrxGroupTx &x, int a0, const TrxqEnt (*q)[kTxRow], const int *nq) {
  const int k = threadIdx.x & (kTxA - 1);
  if (a0 + k < x.S)
    for (int i = threadIdx.x / kTxA; i < nq[k]; i += blockDim.x / kTxA) {
      const TrxqEnt e = q[k][i];
      x.q_fn[(size_t)i * x.S + a0 + k] = e.x;
      x.q_key[(size_t)i * x.S + a0 + k] = e.y;
    }
}

// dgram: n x 154 bytes as they arrived ([0] TN, [1..4] FN big-endian, [5] RSSI, [6..153] one bit per byte); arfcn: n ids (the host
// has checked every header: a call with a bad one queues nothing).
__global__ __launch_bounds__(1024) void k_group_tx_ingest(TrxGroupTx x, int n, const uint8_t *__restrict__ dgram, const int32_t *__restrict__ arfcn,
                                                          TxGainTab gt) {
  __shared__ TrxqEnt q[kTxA][kTxRow];
  __shared__ int32_t lf[kTxWin], lk[kTxWin];                // this round's entries, ARFCN by ARFCN: frame number (then payload slot), key
  __shared__ int16_t fs[kTxWin];                            // the payload slots those entries will be handed, fetched ahead
  __shared__ int32_t cnt[kTxChunks][kTxA];
  __shared__ int nq[kTxA], nf[kTxA], nf0[kTxA], tot[kTxA], lbase[kTxA + 1], st_[kTxA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = blockIdx.x * kTxA;
  TX_STAMP(0, 0);
  if (tid < kTxA) {
    const bool mine = a0 + tid < x.S;
    nq[tid] = mine ? x.q_n[a0 + tid] : 0;
    nf[tid] = mine ? x.free_n[a0 + tid] : 0;
    st_[tid] = 0;
  }
  __syncthreads();
  tx_queues_load(x, a0, q, nq);
  for (int w0 = 0; w0 < n; w0 += kTxWin) {                  // rounds of 4096 datagrams (LDS is sized for one)
    // ---- which of this round's datagrams are ours, and where each goes: counts per (chunk, ARFCN) by ballots ----
    constexpr int CPW = kTxChunks / 16;                     // chunks per wave
    int my_i[CPW], my_k[CPW], my_rank[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; cc++) {
      const int c = wave * CPW + cc;
      const int i = w0 + c * 64 + lane;
      const int local = (i < n ? arfcn[i] : -1) - a0;
      const bool valid = (unsigned)local < (unsigned)kTxA;
      int mycount = 0, rank = 0;
#pragma unroll
      for (int k = 0; k < kTxA; k++) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(valid && local == k);
        if (lane == k) mycount = __builtin_popcountll(m);
        if (local == k) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      }
      if (lane < kTxA) cnt[c][lane] = mycount;
      my_i[cc] = i; my_k[cc] = valid ? local : -1; my_rank[cc] = rank;
    }
    __syncthreads();
    TX_STAMP(0, 1);
    {                                                       // exclusive scan over the chunks, per ARFCN: wave k scans ARFCN k's column
      static_assert(kTxChunks % 64 == 0 && kTxA == 16, "a wave per ARFCN, whole waves of chunks");
      const int k = wave;
      int carry = 0;
      for (int c0 = 0; c0 < kTxChunks; c0 += 64) {
        const int v = cnt[c0 + lane][k];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        cnt[c0 + lane][k] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) { tot[k] = carry; nf0[k] = nf[k]; }
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int k = 0; k < kTxA; k++) { lbase[k] = run; run += tot[k]; }
      lbase[kTxA] = run;
    }
    __syncthreads();
    TX_STAMP(0, 2);
    // ---- every datagram's header into its place (arrival order inside an ARFCN); the payload slots fetched ahead ----
#pragma unroll
    for (int cc = 0; cc < CPW; cc++) {
      if (my_k[cc] < 0) continue;
      const int c = wave * CPW + cc, k = my_k[cc];
      const int lp = lbase[k] + cnt[c][k] + my_rank[cc];
      const uint8_t *d = dgram + (size_t)my_i[cc] * 154;    // (154 i is even: two-byte loads)
      const unsigned h0 = reinterpret_cast<const uint16_t *>(d)[0], h1 = reinterpret_cast<const uint16_t *>(d)[1], h2 = reinterpret_cast<const uint16_t *>(d)[2];
      const int tn = (int)(int8_t)(h0 & 255);
      const unsigned fn = ((h0 >> 8) << 24) | ((h1 & 255) << 16) | ((h1 >> 8) << 8) | (h2 & 255);
      const int RSSI = (int)(int8_t)(h2 >> 8);              // `int RSSI = (int) buffer[5]` on a char buffer (:617)
      const int gi = -RSSI / 10 + 12;                       // scaleVector(*modBurst, pow(10, -RSSI/10)) (:108): integer division
      lf[lp] = (int32_t)fn;
      lk[lp] = (tn & 7) | (gi << 3) | (k << 8) | ((my_i[cc] - w0) << 12);   // (13 bits of position: a round is 8,192 datagrams)
    }
    {
      const int k = tid & (kTxA - 1);
      const int want = tot[k] < nf0[k] ? tot[k] : nf0[k];
      if (a0 + k < x.S)
        for (int j = tid / kTxA; j < want; j += 1024 / kTxA) fs[lbase[k] + j] = x.free_stack[(size_t)(nf0[k] - 1 - j) * x.S + a0 + k];
    }
    __syncthreads();
    TX_STAMP(0, 3);
    // ---- addRadioVector, a lane per ARFCN, everything it touches in LDS ----
    if (tid < kTxA && a0 + tid < x.S) {
      const int k = tid;
      TrxqEnt *row = &q[k][0];
      int n_q = nq[k], n_f = nf[k], used = 0, stt = st_[k];
      for (int e = lbase[k]; e < lbase[k + 1]; e++) {
        int pid = -1;
        if (n_q >= x.qcap || n_f == 0) {                    // queue or payload pool full: the burst is dropped and the ARFCN marked
          stt |= 1;
        } else {
          n_f--;
          pid = fs[lbase[k] + used++];
          n_q = tx_heap_push(row, n_q, trxq_ent(lf[e], (lk[e] & 7) | (pid << 3)));   // mTransmitPriorityQueue.write(newVec) (:109)
        }
        lf[e] = pid;
      }
      nq[k] = n_q; nf[k] = n_f; st_[k] = stt;
    }
    __syncthreads();
    TX_STAMP(0, 4);
    // ---- the payloads to their slots: every thread a word (37 words of bits, then the gain) ----
    const int total = lbase[kTxA] * TRXG_PAYLOAD_WORDS;
    for (int idx = tid; idx < total; idx += 1024) {
      const int e = idx / TRXG_PAYLOAD_WORDS, w = idx - e * TRXG_PAYLOAD_WORDS;
      const int pid = lf[e];
      if (pid < 0) continue;
      const int key = lk[e];
      const int k = (key >> 8) & 15, src = w0 + (key >> 12);
      uint32_t v;
      if (w < 37) {
        const uint16_t *p = reinterpret_cast<const uint16_t *>(dgram + (size_t)src * 154 + 6) + 2 * w;
        v = (uint32_t)p[0] | ((uint32_t)p[1] << 16);        // the bits as they arrive (modulateBurst masks them, sigProcLib.cpp:548)
      } else {
        v = __float_as_uint(gt.v[(key >> 3) & 31]);
      }
      x.pool[((size_t)(a0 + k) * x.npool + pid) * TRXG_PAYLOAD_WORDS + w] = v;
    }
    __syncthreads();                                        // lf / lk / fs / cnt are the next round's
    TX_STAMP(0, 5);
  }
  tx_queues_store(x, a0, q, nq);
  if (tid < kTxA && a0 + tid < x.S) {
    x.q_n[a0 + tid] = nq[tid];
    x.free_n[a0 + tid] = nf[tid];
    if (st_[tid]) x.status[a0 + tid] |= 1u;
  }
  TX_STAMP(0, 6);
}

__device__ __forceinline__ void tx_free(const TrxGroupTx &x, int a, int &nf, int pid) {
  if (pid < 0) return;                                      // the dummy burst is nobody's
  x.free_stack[(size_t)nf * x.S + a] = (int16_t)pid;
  nf++;
}

constexpr int kTxCells = 102 * 8;                           // fillerTable[102][8] (Transceiver.h:79)
// n % m for 0 <= n < 2^22 (a frame number), 1 <= m <= 102 (a filler modulus): the float quotient is off by one at most
__device__ __forceinline__ int tx_fn_mod(int n, int m) {
  int r = n - (int)((float)n * (1.0f / (float)m)) * m;
  r += r < 0 ? m : 0;
  r -= r >= m ? m : 0;
  return r;
}
__global__ __launch_bounds__(256) void k_group_tx_push(TrxGroupTx x, int fn0, int tn0, int n_slots, int16_t *__restrict__ out_pid,
                                                       uint8_t *__restrict__ out_fq) {
  __shared__ TrxqEnt q[kTxA][kTxRow];
  __shared__ int16_t fl[kTxA][kTxCells];                    // the sixteen filler tables
  __shared__ int nq[kTxA], nf[kTxA];
  const int tid = threadIdx.x, a0 = blockIdx.x * kTxA;
  TX_STAMP(1, 0);
  if (tid < kTxA) {
    const bool mine = a0 + tid < x.S;
    nq[tid] = mine ? x.q_n[a0 + tid] : 0;
    nf[tid] = mine ? x.free_n[a0 + tid] : 0;
  }
  __syncthreads();
  tx_queues_load(x, a0, q, nq);
  {
    const int k = tid & (kTxA - 1);
    if (a0 + k < x.S)
      for (int c = tid / kTxA; c < kTxCells; c += 256 / kTxA) fl[k][c] = x.filler[(size_t)c * x.S + a0 + k];
  }
  __syncthreads();
  TX_STAMP(1, 1);
  if (tid < kTxA && a0 + tid < x.S) {
    const int k = tid, a = a0 + tid;
    TrxqEnt *row = &q[k][0];
    int16_t *flk = &fl[k][0];
    int n_q = nq[k], n_f = nf[k];
    int mod[8], r[8];
    int fnc = fn0;                                          // the frame on the air (fn0 < gHyperframe: trxsig_trxgroup_push checks)
#pragma unroll
    for (int m = 0; m < 8; m++) {
      mod[m] = x.fmod[m * x.S + a];
      r[m] = tx_fn_mod(fnc, mod[m]);                        // fnc % fillerModulus[TN], kept up frame by frame
    }
    TrxqEnt top = row[0], c[6];
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = row[1 + i];
    const int t_end = tn0 + n_slots;
    for (int base = 0; base < t_end; base += 8) {           // a frame a turn, its timeslots unrolled (TN is a constant in the body)
#pragma unroll
      for (int tn = 0; tn < 8; tn++) {
        const int tt = base + tn;
        if (tt < tn0 || tt >= t_end) continue;
        const int t = tt - tn0;
        // dump stale bursts, if any: "even if the burst is stale, put it in the filler table" (:142-153)
        while (n_q > 0 && trxq_time_lt(top.x, top.y & 7, fnc, tn)) {
          TrxqEnt e;
          n_q = tx_heap_pop(row, n_q, top, c, e);
          const int etn = e.y & 7;
          int em = mod[0];
#pragma unroll
          for (int m = 1; m < 8; m++) em = etn == m ? mod[m] : em;
          int16_t *cell = &flk[tx_fn_mod(e.x, em) * 8 + etn];
          tx_free(x, a, n_f, *cell);
          *cell = (int16_t)(e.y >> 3);
        }
        int16_t *cell = &flk[r[tn] * 8 + tn];
        int pid = *cell;                                    // the filler entry (:175-177) ...
        int fq = 0;
        if (n_q > 0 && top.x == fnc && (top.y & 7) == tn) {   // ... unless there is data at the desired timestamp (:159-173)
          TrxqEnt e;
          n_q = tx_heap_pop(row, n_q, top, c, e);
          tx_free(x, a, n_f, pid);
          pid = e.y >> 3;
          *cell = (int16_t)pid;
          fq = 1;
        }
        out_pid[(size_t)t * x.S + a] = (int16_t)pid;
        out_fq[(size_t)t * x.S + a] = (uint8_t)fq;
      }
      fnc++;
      const bool wrap = fnc == TRXQ_HYPERFRAME;
      fnc = wrap ? 0 : fnc;
#pragma unroll
      for (int m = 0; m < 8; m++) r[m] = (wrap || r[m] + 1 == mod[m]) ? 0 : r[m] + 1;
    }
    nq[k] = n_q; nf[k] = n_f;
  }
  TX_STAMP(1, 2);
  __syncthreads();
  tx_queues_store(x, a0, q, nq);
  {
    const int k = tid & (kTxA - 1);
    if (a0 + k < x.S)
      for (int c = tid / kTxA; c < kTxCells; c += 256 / kTxA) x.filler[(size_t)c * x.S + a0 + k] = fl[k][c];
  }
  if (tid < kTxA && a0 + tid < x.S) {
    x.q_n[a0 + tid] = nq[tid];
    x.free_n[a0 + tid] = nf[tid];
  }
  TX_STAMP(1, 3);
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

hipError_t trx_launch_group_tx_ingest(hipStream_t st, const TrxGroupTx &x, int n, const uint8_t *dgram, const int32_t *arfcn, const float *gain_tab26) {
  if (n <= 0) return hipSuccess;
  if (x.qcap != kTxQ) return hipErrorInvalidValue;          // (the kernel's LDS copy of a queue)
  TxGainTab gt;
  for (int q = 0; q < 26; q++) gt.v[q] = gain_tab26[q];
  k_group_tx_ingest<<<dim3((x.S + kTxA - 1) / kTxA), dim3(1024), 0, st>>>(x, n, dgram, arfcn, gt);
  return hipGetLastError();
}

hipError_t trx_launch_group_tx_push(hipStream_t st, const TrxGroupTx &x, int fn0, int tn0, int n_slots, int16_t *out_pid, uint8_t *out_fq,
                                    uint8_t *bits_out, float *gain_out, uint8_t *fq_out) {
  if (n_slots <= 0) return hipSuccess;
  if (x.qcap != kTxQ) return hipErrorInvalidValue;
  k_group_tx_push<<<dim3((x.S + kTxA - 1) / kTxA), dim3(256), 0, st>>>(x, fn0, tn0, n_slots, out_pid, out_fq);
  const long long words = (long long)x.S * n_slots * TRXG_PAYLOAD_WORDS;
  k_group_tx_gather<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(x, n_slots, out_pid, out_fq, (uint32_t *)bits_out, gain_out,
                                                                                   fq_out);
  return hipGetLastError();
}
