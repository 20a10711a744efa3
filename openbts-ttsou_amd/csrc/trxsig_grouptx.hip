// trxsig_grouptx.hip -- the TRANSMIT half of the Transceiver group (include/trxsig_trxgroup.h): addRadioVector /
// pushRadioVector (Transceiver/Transceiver.cpp:100-113, 138-181) for S ARFCNs with the priority queue, the stale-burst dump
// and the filler table [FN % modulus][TN] on the device.
//   k_group_tx_arrive : driveTransmitPriorityQueue's parsing (:596-620) ON THE DEVICE (round 5) from the raw 154-byte datagrams and
//                       their ARFCN ids, as they arrived -- everything that needs no queue state, so it runs on the uploads' stream: a
//                       workgroup owns sixteen ARFCNs, finds its datagrams (a stable counting sort by wave ballots: arrival order is
//                       kept inside an ARFCN) and parses TN / big-endian FN / RSSI into per-ARFCN lists;
//   k_group_tx<INGEST, WALK> : the queues' kernel, a workgroup of four waves for four ARFCNs, in three forms -- the add call's part
//                       alone, the push's alone, or both in one launch (the add call's part stays pending until the push that follows):
//     INGEST            addRadioVector (:100-113) for those lists: WAVE k enters ARFCN k's bursts in its queue -- the queue sits in LDS
//                       for the duration, one word an entry (trxsig_txq_lds.h: std::priority_queue's moves, comparisons as integer
//                       comparisons), the payload slots to hand out fetched ahead -- while a thread per burst copies the payload (148
//                       bits + gain) to its slot;
//     WALK              pushRadioVector: WAVE k walks ARFCN k's n_slots timeslots, every value uniform across its lanes (scalar
//                       branches): stale entries leave the queue for the filler table, the entry for exactly this time (if any)
//                       replaces the filler entry and goes out, else the filler entry goes out (:142-177) -- as payload REFERENCES,
//                       nothing is copied on the serial path; queues AND filler tables are in LDS for the walk (round 4: a dependent
//                       global access per queue move and per slot was the whole kernel); a queue of at most 64 entries is popped
//                       ACROSS the wave's lanes (tx_lane_pop); at the end of every turn of the walk the referenced payloads are copied
//                       into the layout trxsig_txbe_push_bursts takes ([S][n][148] bits, [S][n] gains): what the fused transmit back
//                       end then modulates, resamples and packs to int16.
// A queue holding a burst 2^17 frames or more from the call's own frame cannot be said in one-word entries: its workgroup works on the
// arrays in memory with trxsig_txq.h's moves instead (the slow path: same results).
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
#define TX_ACC_BEGIN() const unsigned long long acc_t0_ = clock64()
#define TX_ACC_END(kern, i) do { acc_[i - 4] += clock64() - acc_t0_; } while (0)
#define TX_ACC_DECL() unsigned long long acc_[4] = {0, 0, 0, 0}
#define TX_ACC_OUT(kern) do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 4; i_++) g_txprobe[kern][4 + i_] = acc_[i_]; } while (0)
extern "C" int trx_txprobe_read(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_txprobe), sizeof(g_txprobe)); }
#else
#define TX_STAMP(kern, i) do { } while (0)
#define TX_ACC_BEGIN() do { } while (0)
#define TX_ACC_END(kern, i) do { } while (0)
#define TX_ACC_DECL() do { } while (0)
#define TX_ACC_OUT(kern) do { } while (0)
#endif

namespace {

constexpr int kTxA = 16;                                   // ARFCNs per workgroup
constexpr int kTxQ = TRXQ_LDS_CAP;                          // queue entries per ARFCN held in LDS (= the queue's capacity, trxsig_trxgroup.cpp)
constexpr int kTxRow = kTxQ + 1;                            // an ARFCN's row in LDS: one word of padding (the rows start on different banks;
                                                            // speculative reads past the queue's end land on it)
constexpr int kTxWin = 8192;                                // datagrams per round: 128 chunks of a wave's width
constexpr int kTxChunks = kTxWin / 64;

struct TxGainTab { float v[26]; };                          // pow(10, q), q = -12 .. 13 (host: the reference's double pow, rounded to float)

// the sixteen queues of a workgroup, memory -> LDS, packed relative to frame `ref` (trxsig_txq_lds.h); *far is set when an entry
// lies outside the packed form's window (the workgroup then works on the arrays in memory: the slow path)
template <int A>
__device__ __forceinline__ void tx_queues_load(const TrxGroupTx &x, int a0, TrxqPk (*q)[kTxRow], const int *nq, int ref, int *far) {
  const int k = threadIdx.x & (A - 1);
  if (a0 + k < x.S)
    for (int i = threadIdx.x / A; i < nq[k]; i += blockDim.x / A) {
      const int32_t fn = x.q_fn[(size_t)i * x.S + a0 + k], key = x.q_key[(size_t)i * x.S + a0 + k];
      if (!trxq_pk_ok(fn, ref) || (key >> 3) >= TRXQ_PK_IDS) *far = 1;
      q[k][i] = trxq_pk(fn, key & 7, (key >> 3) & (TRXQ_PK_IDS - 1), ref);
    }
}
template <int A>
__device__ __forceinline__ void tx_queues_store(const TrxGroupTx &x, int a0, const TrxqPk (*q)[kTxRow], const int *nq, int ref) {
  const int k = threadIdx.x & (A - 1);
  if (a0 + k < x.S)
    for (int i = threadIdx.x / A; i < nq[k]; i += blockDim.x / A) {
      const TrxqPk e = q[k][i];
      x.q_fn[(size_t)i * x.S + a0 + k] = trxq_pk_fn(e, ref);
      x.q_key[(size_t)i * x.S + a0 + k] = trxq_pk_tn(e) | (trxq_pk_id(e) << 3);
    }
}

// ---- the arrival half of an add call (no queue state in it: it runs on the uploads' stream, beside the previous batch's walk) ----
// dgram: n x 154 bytes as they arrived ([0] TN, [1..4] FN big-endian, [5] RSSI, [6..153] one bit per byte); arfcn: n ids (the host
// has checked every header: a call with a bad one queues nothing).  A workgroup owns sixteen ARFCNs; per round of 8,192 datagrams it
// leaves, in memory: tot[16] (how many datagrams each of its ARFCNs got) and the round's entries ARFCN by ARFCN, arrival order kept
// inside an ARFCN -- lf (frame number), lk (tn | gain index << 3 | local ARFCN << 8 | position in the round << 12).
struct TxArrive {
  int32_t *lf, *lk;                                         // [workgroups][n_pad]
  int32_t *tot;                                             // [workgroups][rounds][16]
  int n_pad, rounds;                                        // n_pad = rounds * kTxWin
};
__global__ __launch_bounds__(1024) void k_group_tx_arrive(int S, int n, const uint8_t *__restrict__ dgram, const int32_t *__restrict__ arfcn, TxArrive ar) {
  __shared__ int32_t cnt[kTxChunks][kTxA];
  __shared__ int tot[kTxA], lbase[kTxA + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = blockIdx.x * kTxA;
  int32_t *const olf = ar.lf + (size_t)blockIdx.x * ar.n_pad, *const olk = ar.lk + (size_t)blockIdx.x * ar.n_pad;
  int round = 0;
  for (int w0 = 0; w0 < n; w0 += kTxWin, round++) {         // rounds of 8,192 datagrams (the ingest kernel's LDS is sized for one)
    // ---- which of this round's datagrams are ours, and where each goes: counts per (chunk, ARFCN), ranks inside a chunk ----
    constexpr int CPW = kTxChunks / 16;                     // chunks per wave
    int my_i[CPW], my_k[CPW], my_rank[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; cc++) {
      const int c = wave * CPW + cc;
      const int i = w0 + c * 64 + lane;
      const int local = (i < n ? arfcn[i] : -1) - a0;
      const bool valid = (unsigned)local < (unsigned)kTxA;
      // the lanes holding the same ARFCN as this one: four ballots, one per bit of the local id
      unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((local >> b) & 1);
        peers &= ((local >> b) & 1) ? m : ~m;
      }
      const int rank = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
      if (lane < kTxA) cnt[c][lane] = 0;
      if (valid && rank == 0) cnt[c][local] = __builtin_popcountll(peers);   // (same wave, after the zeros: LDS keeps a wave's order)
      my_i[cc] = i; my_k[cc] = valid ? local : -1; my_rank[cc] = rank;
    }
    __syncthreads();
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
      if (lane == 0) {
        tot[k] = carry;
        ar.tot[((size_t)blockIdx.x * ar.rounds + round) * kTxA + k] = a0 + k < S ? carry : 0;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int k = 0; k < kTxA; k++) { lbase[k] = run; run += tot[k]; }
      lbase[kTxA] = run;
    }
    __syncthreads();
    // ---- every datagram's header into its place (arrival order inside an ARFCN) ----
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
      olf[w0 + lp] = (int32_t)fn;
      olk[w0 + lp] = (tn & 7) | (gi << 3) | (k << 8) | ((my_i[cc] - w0) << 12);   // (13 bits of position: a round is 8,192 datagrams)
    }
    __syncthreads();                                        // cnt / tot / lbase are the next round's
  }
}

// ---- the queue half (the queues' stream): addRadioVector for what k_group_tx_arrive sorted, the payloads to their slots.
// ref: the frame the packed queue entries are relative to (the first datagram's); far != 0: the host saw a datagram outside the packed
// window (every workgroup takes the slow path).
__device__ __forceinline__ void tx_free(const TrxGroupTx &x, int a, int &nf, int pid) {
  if (pid < 0) return;                                      // the dummy burst is nobody's
  x.free_stack[(size_t)nf * x.S + a] = (int16_t)pid;
  nf++;
}

constexpr int kTxCells = 102 * 8;                           // fillerTable[102][8] (Transceiver.h:79)
// n % m for 0 <= n < 2^22 (a frame number), 1 <= m <= 102 (a filler modulus), rm = (float)(1 / m): the float quotient is off by
// one at most
__device__ __forceinline__ int tx_fn_mod(int n, int m, float rm) {
  int r = n - (int)((float)n * rm) * m;
  r += r < 0 ? m : 0;
  r -= r >= m ? m : 0;
  return r;
}
// ---- std::pop_heap on a queue of at most 64 entries held ACROSS THE WAVE'S LANES (entry i in lane i of hv), every lane working on the
// one pop.  A lone wave issues an instruction every four clocks at best, whichever unit executes it: the walk's time is its instruction
// count (profiles/r05_group_tx_probe.txt: ~330 instructions, ~1,300 clocks a pop with the queue in LDS).  Here
//   pref  bit c (c >= 1): the hole, arriving at c's parent, goes to c -- for the right child (even c) unless comp(first[c], first[c - 1]),
//         for the left child (odd c) if comp(first[c + 1], first[c]): one vector compare for all the pairs (the neighbours' entries and
//         answers by DPP wave shifts) and a ballot.  The pairs the hole meets are untouched by the pop so far, so the answers taken
//         before the descent are the ones __adjust_heap would form;
//   path  the hole's way down = the chain of preferred nodes from the root: lane c is on it when c and all its ancestors but the root
//         are preferred (their positions: a constant of the lane) -- a second ballot -- cut where the descent stops (hole < (len - 1) / 2
//         fails), then the lone left child (len even).  A level's position exceeds the level's above: the path is a 64-bit SET;
//   G     bit i: comp(first[i], value) for the queue's last element `value`.  __push_heap's climb back stops at the deepest path
//         position (or the root) whose element is not later than value: the highest bit of path & ~G;
//   the elements above that position move up one level each (lane pos[l] takes lane pos[l + 1]'s entry: one ds_bpermute), value lands
//   there, the levels below keep what they had.
// The same comparisons on the same elements as trxq_pop / tx_heap_pop, the same array afterwards (the tests hold the kernels against
// std::priority_queue: ties, queues up to 64 deep here, deeper ones on the LDS form).  ~60 instructions and no loop.  n = size before
// (1 .. 64); top in / out as in tx_heap_pop; anc: the lane's own position and its ancestors' but the root's.
__device__ __forceinline__ int tx_lane_pop(TrxqPk &hv, int n, TrxqPk &top, TrxqPk &popped, unsigned long long anc) {
  const int lane = (int)(threadIdx.x & 63);
  popped = top;
  const int len = n - 1;
  if (len == 0) return 0;
  const TrxqPk v = __builtin_amdgcn_readlane(hv, len);
  const int half = (len - 1) >> 1;
  const int lone = (len & 1) ? -1 : (len - 2) >> 1;
  // which child the hole would take at every node at once: lane c (c >= 1) is PREFERRED when the hole, arriving at c's parent, goes to
  // c -- the right child (even c) unless comp(first[c], first[c - 1]), the left child (odd c) if comp(first[c + 1], first[c])
  const TrxqPk left = __builtin_amdgcn_update_dpp(0, hv, 0x138, 0xf, 0xf, false);    // wave_shr:1 -- lane c: the entry of lane c - 1
  const int later = tx_gt(hv, left) ? 1 : 0;                                          // comp(first[c], first[c - 1])
  const int later_r = __builtin_amdgcn_update_dpp(0, later, 0x130, 0xf, 0xf, false);  // wave_shl:1 -- lane c: that of lane c + 1
  const unsigned long long pref = __builtin_amdgcn_ballot_w64((lane & 1) ? later_r != 0 : later == 0);
  const unsigned long long G = __builtin_amdgcn_ballot_w64(tx_gt(hv, v));
  // the hole's way down is the chain of preferred nodes from the root: lane c is on it when c and all its ancestors but the root are
  // preferred (anc: their positions, a constant of the lane) -- cut where the descent stops, at the first hole without two children
  // (a node is reached when its parent is < half, i.e. when its position is <= 2 half)
  unsigned long long path = __builtin_amdgcn_ballot_w64(lane >= 1 && (pref & anc) == anc) & ((2ull << (2 * half)) - 1ull);
  int hole = path ? 63 - __builtin_clzll(path) : 0, lonepos = -1;
  if (hole == lone) {
    lonepos = hole;
    hole = 2 * hole + 1;
    path |= 1ull << hole;
  }
  // the climb back stops at the deepest level of the path (the root included) whose element is NOT later than value
  const unsigned long long stop = (path & ~G) | 1ull;
  const int posj = 63 - __builtin_clzll(stop);
  const unsigned long long moved = (path | 1ull) & ((1ull << posj) - 1ull);   // the levels above it take their chosen child's entry
  const int sh = 2 * lane + 2;
  const int right = sh < 64 ? (int)((pref >> sh) & 1ull) : 0;                 // this node's right child is the preferred one
  const int child = lane == lonepos ? 2 * lane + 1 : 2 * lane + 1 + right;
  const int src = ((moved >> lane) & 1ull) ? child : lane;
  TrxqPk hn = __builtin_amdgcn_ds_bpermute(src << 2, hv);
  hn = lane == posj ? v : hn;
  hv = hn;
  top = __builtin_amdgcn_readlane(hv, 0);
  return len;
}

// ---- std::push_heap the same way.  The path from the new leaf (position n) to the root is the set A of its ancestors: lane i is one
// exactly when i + 1 is a proper binary prefix of n + 1 -- one shift and a compare a lane, one ballot.  G bit i = comp(first[i], value).
// __push_heap climbs while the parent is later than value: it stops under the deepest ancestor that is NOT later (the highest bit of
// A & ~G), or at the root; the ancestors below that point move down one level along the path (a path position takes its parent's entry:
// one bpermute), value lands in the gap.  n = size before (0 .. 63); returns n + 1.  bl = 32 - clz(lane + 1), par = (lane - 1) >> 1.
__device__ __forceinline__ int tx_lane_push(TrxqPk &hv, int n, TrxqPk v, int bl, int par) {
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long G = __builtin_amdgcn_ballot_w64(tx_gt(hv, v));
  const int m1 = n + 1, sh = (32 - __builtin_clz(m1)) - bl;
  const unsigned long long A = __builtin_amdgcn_ballot_w64(sh > 0 && (m1 >> sh) == lane + 1);
  const unsigned long long pathset = A | (1ull << n);
  const unsigned long long stop = A & ~G;
  const unsigned long long upto = stop ? (2ull << (63 - __builtin_clzll(stop))) - 1ull : 0ull;   // the positions up to the stopping ancestor
  const unsigned long long below = pathset & ~upto;         // the path under it: its lowest position takes value, the others their parent's entry
  const int land = __builtin_ctzll(below);
  const unsigned long long takers = below & (below - 1ull);
  const int src = ((takers >> lane) & 1ull) ? par : lane;
  const TrxqPk hn = __builtin_amdgcn_ds_bpermute(src << 2, hv);
  hv = lane == land ? v : hn;
  return n + 1;
}

constexpr int kTxI = 4;                                     // ARFCNs per workgroup of the queues' kernel: a WAVE each, a SIMD each (sixteen waves of
                                                            // this scalar code on one CU took ~1.6 times as long per push, and the workgroup
                                                            // waited for the slowest of sixteen)
constexpr int kTxWalk = 128;                                // timeslots a turn of the walk (their filler cells are worked out ahead, by every thread)
constexpr int kTxP = kTxI;
// The walk of an ARFCN's queue is one thread's work, a chain of dependent instructions; what it costs is the instructions the wave
// issues (~8 cycles each with nothing to hide them behind).  A wave per ARFCN, every value the same in all its lanes
// (readfirstlane'd ARFCN index, LDS addresses that do not depend on the lane): the branches are scalar branches, no execution
// masks to save, combine and restore as with a lane per ARFCN (where every branch some lane takes all sixteen pay for).
// far_in != 0: the walk is too long for the packed form's window (the host's check)
// ONE kernel for the queues' stream, in three forms:
//   <true, false>  addRadioVector for what k_group_tx_arrive sorted (ref: the frame the packed queue entries are relative to -- the
//                  first datagram's);
//   <false, true>  pushRadioVector for n_slots timeslots from (fn0, tn0);
//   <true, true>   both, the add call's ingest left pending until the push that follows it (trxsig_trxgroup.cpp: the usual order of a
//                  transmit loop): the queues go to LDS once, are entered into and walked, and go back once -- a dependent launch
//                  and a round trip of the queues less on the serial chain.  The packed entries are relative to fn0 then.
// far_in != 0: some datagram lies outside the packed entries' window round the reference frame (the host's check), or the walk is
// too long for it: every workgroup works on the arrays in memory (trxsig_txq.h's moves; same results, slow).
template <bool INGEST, bool WALK>
__global__ __launch_bounds__(64 * kTxI) void k_group_tx(TrxGroupTx x, int n, const uint8_t *__restrict__ dgram, TxArrive ar, TxGainTab gt, int ref_in,
                                                        int fn0, int tn0, int n_slots, uint32_t *__restrict__ bits_out, float *__restrict__ gain_out,
                                                        uint8_t *__restrict__ fq_out, int far_in) {
  static_assert(kTxA % kTxI == 0, "a workgroup's ARFCNs lie inside one arrival workgroup's sixteen");
  constexpr int NT = 64 * kTxI;
  __shared__ TrxqPk q[kTxI][kTxRow];                        // the queues, packed relative to `ref`
  __shared__ int nq[kTxI], nf[kTxI], st_[kTxI], far;
  // the ingest's
  __shared__ int32_t lf[INGEST ? kTxWin + 1 : 1], lk[INGEST ? kTxWin + 1 : 1];   // this round's entries, ARFCN by ARFCN: frame number, key
  __shared__ int16_t fs[INGEST ? kTxWin + 2 : 2];           // the payload slots those entries are handed, fetched ahead
  __shared__ int tot[kTxI], acc[kTxI], lbase[kTxI + 1], gbase;
  // the walk's
  __shared__ int16_t fl[WALK ? kTxP : 1][kTxCells];         // the filler tables
  __shared__ uint16_t cidx[kTxP][kTxWalk + 2];              // [FN % modulus][TN] of the slots of this turn (+ the next turn's first)
  __shared__ int16_t opid[kTxP][kTxWalk];                   // what goes out at each slot of the turn: a payload reference (-1: the dummy burst) ...
  __shared__ uint8_t ofq[kTxP][kTxWalk];                    // ... and whether it came from the queue
  __shared__ int md[kTxP][8];
  __shared__ float mdr[kTxP][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = blockIdx.x * kTxI;
  const int ref = WALK ? fn0 : ref_in;                      // (the walk's slot times are tn0 + t when the entries are relative to fn0)
  TX_STAMP(WALK ? 1 : 0, 0);
  if (tid < kTxI) {
    const bool mine = a0 + tid < x.S;
    nq[tid] = mine ? x.q_n[a0 + tid] : 0;
    nf[tid] = mine ? x.free_n[a0 + tid] : 0;
    st_[tid] = 0;
  }
  if (WALK && tid < kTxP * 8) {
    const int k = tid & (kTxP - 1), m = tid / kTxP;
    const int v = a0 + k < x.S ? x.fmod[m * x.S + a0 + k] : 1;
    md[k][m] = v;
    mdr[k][m] = 1.0f / (float)v;
  }
  if (tid == 0) far = far_in;
  __syncthreads();
  tx_queues_load<kTxI>(x, a0, q, nq, ref, &far);
  if (WALK) {
    const int k = tid & (kTxP - 1);
    if (a0 + k < x.S)
      for (int c = tid / kTxP; c < kTxCells; c += NT / kTxP) fl[k][c] = x.filler[(size_t)c * x.S + a0 + k];
  }
  if (INGEST) {
    // ================= addRadioVector for what k_group_tx_arrive sorted =================
    const int g16 = a0 / kTxA, k0 = a0 - g16 * kTxA;        // the arrival workgroup whose lists hold this workgroup's ARFCNs, and where in its sixteen
    const int32_t *const ilf = ar.lf + (size_t)g16 * ar.n_pad, *const ilk = ar.lk + (size_t)g16 * ar.n_pad;
    int round = 0;
    for (int w0 = 0; w0 < n; w0 += kTxWin, round++) {         // rounds of 8,192 datagrams (LDS is sized for one)
      if (tid == 0) {                                         // this workgroup's part of the arrival workgroup's lists
        const int32_t *t16 = ar.tot + ((size_t)g16 * ar.rounds + round) * kTxA;
        int run = 0;
        for (int k = 0; k < k0; k++) run += t16[k];
        gbase = run;
        int loc = 0;
        for (int k = 0; k < kTxI; k++) {
          const int t = t16[k0 + k];
          tot[k] = t; lbase[k] = loc; loc += t;
          // the queue and the payload pool only fill up during a call: what is accepted is a PREFIX of the ARFCN's arrivals
          // ("if the queue or the pool is full the burst is dropped and the ARFCN marked")
          const int room = min(x.qcap - nq[k], nf[k]);
          acc[k] = t < room ? t : (room > 0 ? room : 0);
        }
        lbase[kTxI] = loc;
      }
      __syncthreads();
      TX_STAMP(0, 2);
      for (int e = tid; e < lbase[kTxI]; e += NT) { lf[e] = ilf[w0 + gbase + e]; lk[e] = ilk[w0 + gbase + e]; }
      {                                                       // the payload slots fetched ahead
        const int k = tid & (kTxI - 1);
        if (a0 + k < x.S)
          for (int j = tid / kTxI; j < acc[k]; j += NT / kTxI) fs[lbase[k] + j] = x.free_stack[(size_t)(nf[k] - 1 - j) * x.S + a0 + k];
      }
      __syncthreads();
      TX_STAMP(0, 3);
      {
        // ---- the payloads to their slots, a burst a thread: its 148 bytes as aligned sixteen-byte loads, all in flight at once; the
        //      thread's FIRST burst is loaded before the queue insertions below and stored after them (the loads land meanwhile) ----
        // the bits as they arrive (modulateBurst masks them, sigProcLib.cpp:548).  Datagram src's payload starts at byte 154 src + 6: on a
        // multiple of four for odd src, two past one for even src -- aligned words are loaded and shifted by two bytes then
        uint32_t aw[38];
        auto load = [&](int e, bool &odd2, long long &slot, uint32_t &gain) {
          slot = -1; odd2 = false; gain = 0;
          if (e >= lbase[kTxI]) return;
          const int key = lk[e];
          const int kk = ((key >> 8) & 15) - k0, src = w0 + (key >> 12);
          if (e - lbase[kk] >= acc[kk]) return;               // dropped
          const size_t pb = (size_t)src * 154 + 6;
          odd2 = (pb & 2) != 0;
          const uint32_t *pa = reinterpret_cast<const uint32_t *>(dgram + (pb & ~(size_t)3));
  #pragma unroll
          for (int w = 0; w < 36; w += 4) __builtin_memcpy(&aw[w], pa + w, 16);
          __builtin_memcpy(&aw[36], pa + 36, 8);              // (up to four bytes past the last datagram's end: the array is allocated eight longer)
          gain = __float_as_uint(gt.v[(key >> 3) & 31]);
          slot = ((long long)(a0 + kk) * x.npool + fs[e]) * TRXG_PAYLOAD_WORDS;
        };
        auto store = [&](bool odd2, long long slot, uint32_t gain) {
          if (slot < 0) return;
          uint32_t v[TRXG_PAYLOAD_WORDS];
  #pragma unroll
          for (int w = 0; w < 37; w++) v[w] = odd2 ? __builtin_amdgcn_alignbyte(aw[w + 1], aw[w], 2) : aw[w];
          v[37] = gain;
          uint32_t *dst = x.pool + slot;
          static_assert(TRXG_PAYLOAD_WORDS == 38, "nine sixteen-byte stores and an eight-byte one");
  #pragma unroll
          for (int w = 0; w < 36; w += 4) __builtin_memcpy(dst + w, &v[w], 16);   // (a slot starts on a multiple of 8 bytes)
          __builtin_memcpy(dst + 36, &v[36], 8);
        };
        bool odd2;
        long long slot;
        uint32_t gain;
        load(tid, odd2, slot, gain);
        // ---- addRadioVector: WAVE k enters ARFCN k's bursts (every value the same in all its lanes: scalar branches, see k_group_tx_push;
        //      lane 0's stores count) ----
        const int k = __builtin_amdgcn_readfirstlane(wave);
        if (a0 + k < x.S) {
          const int e0 = lbase[k], m = acc[k];
          int n_q = nq[k];
          if (!far && n_q + m <= 64) {
            // the queue fits the wave's lanes, this round's bursts included: entry i in lane i, the round's new entries packed side by
            // side in another register (lane j the j-th), and every push the work of all lanes (tx_lane_push)
            TrxqPk *row = &q[k][0];
            TrxqPk hv = row[lane];                            // (entries past the queue's end: whatever the row holds, never looked at)
            const int e = e0 + (lane < m ? lane : 0);
            const TrxqPk vpk = trxq_pk(lf[e], lk[e] & 7, fs[e], ref);
            const int bl = 32 - __builtin_clz(lane + 1), par = (lane - 1) >> 1;
            for (int j = 0; j < m; j++) n_q = tx_lane_push(hv, n_q, __builtin_amdgcn_readlane(vpk, j), bl, par);   // mTransmitPriorityQueue.write(newVec) (:109)
            if (lane < n_q) row[lane] = hv;
          } else if (!far) {
            TrxqPk *row = &q[k][0];
            int32_t f1 = lf[e0], k1 = lk[e0];
            int s1 = fs[e0];
            for (int j = 0; j < m; j++) {                     // the same with the queue in LDS; the next entry fetched meanwhile
              const TrxqPk v = trxq_pk(f1, k1 & 7, s1, ref);
              f1 = lf[e0 + j + 1]; k1 = lk[e0 + j + 1]; s1 = fs[e0 + j + 1];   // (one past the ARFCN's last: the next ARFCN's or padding, unused)
              n_q = tx_heap_push(row, n_q, v);
            }
          } else if (lane == 0) {
            const TrxqView gq = {x.q_fn + a0 + k, x.q_key + a0 + k, x.S};
            for (int j = 0; j < m; j++) n_q = trxq_push(gq, n_q, lf[e0 + j], (lk[e0 + j] & 7) | ((int)fs[e0 + j] << 3));
          }
          n_q = __builtin_amdgcn_readfirstlane(n_q);
          if (lane == 0) {                                    // (nq / nf / st_ are not read by the copy)
            nq[k] = n_q; nf[k] -= m;
            if (m < tot[k]) st_[k] |= 1;                      // queue or payload pool full: the rest is dropped and the ARFCN marked
          }
        }
        TX_STAMP(0, 4);
        store(odd2, slot, gain);
        for (int e = tid + NT; e < lbase[kTxI]; e += NT) {    // (acc / lbase / fs / lk are not written above)
          load(e, odd2, slot, gain);
          store(odd2, slot, gain);
        }
      }
      __syncthreads();                                        // lf / lk / fs are the next round's
      TX_STAMP(0, 5);
    }

  }
  __syncthreads();
  TX_STAMP(1, 1);
  if (WALK) {
    // ================= pushRadioVector for n_slots timeslots =================
    // wave k walks ARFCN a0 + k; what it keeps between the turns (the same in every lane):
    const int k = __builtin_amdgcn_readfirstlane(tid >> 6), a = a0 + k;
    const bool walker = a < x.S;
    const bool writer = (tid & 63) == 0;                      // (one lane stores; all of them compute)
    TrxqPk *row = &q[k][0];
    int16_t *flk = &fl[k][0];
    int n_q = nq[k], n_f = nf[k];
    const int is_far = far;
    TrxqPk top = 0, last = 0, c[6] = {0, 0, 0, 0, 0, 0};
    // a queue of at most 64 entries is walked across the wave's lanes (tx_lane_pop), a longer one in LDS (tx_heap_pop): a walk only pops
    const bool lanes = walker && !is_far && n_q <= 64;
    TrxqPk hv = 0;
    unsigned long long anc = 0;                               // this lane's position and its ancestors' but the root's (tx_lane_pop)
    for (int p = tid & 63; p > 0; p = (p - 1) >> 1) anc |= 1ull << p;
    if (lanes) {
      hv = row[tid & 63];                                     // (entries past the queue's end: whatever the row holds, never looked at)
      top = __builtin_amdgcn_readlane(hv, 0);
    } else if (walker && !is_far && n_q > 0) {
      top = row[0]; last = row[n_q - 1];
  #pragma unroll
      for (int i = 0; i < 6; i++) c[i] = row[1 + i];
    }
    const TrxqView gq = {x.q_fn + a, x.q_key + a, x.S};       // (the slow path: the queue where it lives)
    TX_ACC_DECL();
    for (int t0 = 0; t0 < n_slots; t0 += kTxWalk) {
      const int nt = min(kTxWalk, n_slots - t0);
      for (int i = tid; i < kTxP * (nt + 1); i += NT) {       // the filler cell of every slot of the turn: [FN % modulus][TN]
        const int kk = i & (kTxP - 1), j = i / kTxP, now = tn0 + t0 + j, tn = now & 7;
        int fn = fn0 + (now >> 3);                            // (fn0 < gHyperframe, n_slots < 8 gHyperframe: trxsig_trxgroup_push checks)
        fn -= fn >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
        fn -= fn >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
        cidx[kk][j] = (uint16_t)(tx_fn_mod(fn, md[kk][tn], mdr[kk][tn]) * 8 + tn);
      }
      __syncthreads();
      if (walker && !is_far) {
        int cell = cidx[k][0];
        int pid = flk[cell];                                  // the filler entry of the slot (:175-177), read ahead
        for (int j = 0; j < nt; j++) {
          const int t = t0 + j, now = tn0 + t;                // (packed relative to fn0: the time of this slot IS tn0 + t)
          const int cell_next = cidx[k][j + 1];
          int fq = 0;
          bool reread = false;
          while (n_q > 0) {
            const int tk = trxq_pk_time(top);
            if (tk > now) break;
            TrxqPk e;
            { TX_ACC_BEGIN(); n_q = lanes ? tx_lane_pop(hv, n_q, top, e, anc) : tx_heap_pop(row, n_q, top, c, last, e); TX_ACC_END(1, 4); }
            if (tk == now) {                                  // the burst for exactly this slot (:159-173): it replaces the filler entry and goes out
              const int old = reread ? (int)flk[cell] : pid;
              if (old >= 0) { if (writer) x.free_stack[(size_t)n_f * x.S + a] = (int16_t)old; n_f++; }
              pid = trxq_pk_id(e);
              flk[cell] = (int16_t)pid;
              reread = false;
              fq = 1;
              break;
            }
            // a stale burst: "even if the burst is stale, put it in the filler table" (:142-153), [FN % modulus][TN] of ITS time
            const int etn = trxq_pk_tn(e);
            const int ecell = tx_fn_mod(trxq_pk_fn(e, fn0), md[k][etn], mdr[k][etn]) * 8 + etn;
            const int old = flk[ecell];
            if (old >= 0) { if (writer) x.free_stack[(size_t)n_f * x.S + a] = (int16_t)old; n_f++; }
            flk[ecell] = (int16_t)trxq_pk_id(e);
            reread = reread || ecell == cell;
          }
          if (reread) pid = flk[cell];
          if (writer) { opid[k][j] = (int16_t)pid; ofq[k][j] = (uint8_t)fq; }
          cell = cell_next;
          pid = flk[cell];                                    // (after this slot's writes: LDS keeps a wave's order)
        }
      } else if (walker) {
        // the slow path: trxsig_txq.h's moves on the arrays in memory, a dependent access a move (lane 0 alone: the moves store)
        if (writer)
          for (int j = 0; j < nt; j++) {
            const int t = t0 + j, tn = (tn0 + t) & 7;
            int fnc = fn0 + ((tn0 + t) >> 3);
            fnc -= fnc >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
            fnc -= fnc >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
            int fq = 0;
            while (n_q > 0) {
              const int32_t tfn = gq.f(0), tkey = gq.k(0);
              const bool stale = trxq_time_lt(tfn, tkey & 7, fnc, tn), hit = tfn == fnc && (tkey & 7) == tn;
              if (!stale && !hit) break;
              int32_t efn, ekey;
              n_q = trxq_pop(gq, n_q, &efn, &ekey);
              const int etn = ekey & 7;
              int16_t *cl = &flk[(efn % md[k][etn]) * 8 + etn];
              tx_free(x, a, n_f, *cl);
              *cl = (int16_t)(ekey >> 3);
              if (!stale) { fq = 1; break; }
            }
            opid[k][j] = flk[cidx[k][j]];
            ofq[k][j] = (uint8_t)fq;
          }
        n_q = __builtin_amdgcn_readfirstlane(n_q); n_f = __builtin_amdgcn_readfirstlane(n_f);
      }
      __syncthreads();
      // ---- the turn's output: the referenced payloads into the layout trxsig_txbe_push_bursts takes -- bits_out [S][n_slots][148],
      //      gain_out / fq_out [S][n_slots] -- a (slot, ARFCN) cell a thread, its 152 bytes in flight at once.  (A separate kernel until
      //      round 5: a launch more on the chain the next batch's ingest waits for.) ----
      for (int i = tid; i < kTxP * nt; i += NT) {
        const int kk = i / nt, j = i - kk * nt, aa = a0 + kk;
        if (aa >= x.S) continue;
        const int pid = opid[kk][j];
        const uint32_t *src = pid < 0 ? x.dummy : x.pool + ((size_t)aa * x.npool + pid) * TRXG_PAYLOAD_WORDS;
        uint32_t v[TRXG_PAYLOAD_WORDS];
  #pragma unroll
        for (int w = 0; w < 36; w += 4) __builtin_memcpy(&v[w], src + w, 16);
        __builtin_memcpy(&v[36], src + 36, 8);
        const size_t cell = (size_t)aa * n_slots + (t0 + j);
        uint32_t *dst = bits_out + cell * 37;
  #pragma unroll
        for (int w = 0; w < 36; w += 4) __builtin_memcpy(dst + w, &v[w], 16);
        dst[36] = v[36];
        gain_out[cell] = __uint_as_float(v[37]);
        fq_out[cell] = ofq[kk][j];
      }
    }
    if (lanes && (int)(tid & 63) < n_q) row[tid & 63] = hv;   // the queue back to its row
    if (walker && writer) { nq[k] = n_q; nf[k] = n_f; }

    TX_ACC_OUT(1);
  }
  TX_STAMP(1, 2);
  __syncthreads();
  if (!far) tx_queues_store<kTxI>(x, a0, q, nq, ref);
  if (WALK) {
    const int kk = tid & (kTxP - 1);
    if (a0 + kk < x.S)
      for (int c2 = tid / kTxP; c2 < kTxCells; c2 += NT / kTxP) x.filler[(size_t)c2 * x.S + a0 + kk] = fl[kk][c2];
  }
  if (tid < kTxI && a0 + tid < x.S) {
    x.q_n[a0 + tid] = nq[tid];
    x.free_n[a0 + tid] = nf[tid];
    if (INGEST && st_[tid]) x.status[a0 + tid] |= 1u;
  }
  TX_STAMP(WALK ? 1 : 0, WALK ? 3 : 6);
}

}  // namespace

size_t trx_group_tx_arrive_ints(int S, int n, size_t *tot_ints) {
  const size_t nblk = (size_t)(S + kTxA - 1) / kTxA, rounds = ((size_t)n + kTxWin - 1) / kTxWin;
  if (tot_ints) *tot_ints = nblk * rounds * kTxA;
  return nblk * rounds * kTxWin;
}

static TxArrive tx_arrive_args(int n, int32_t *a_lf, int32_t *a_lk, int32_t *a_tot) {
  TxArrive ar;
  ar.lf = a_lf; ar.lk = a_lk; ar.tot = a_tot;
  ar.rounds = (n + kTxWin - 1) / kTxWin;
  ar.n_pad = ar.rounds * kTxWin;
  return ar;
}

hipError_t trx_launch_group_tx_arrive(hipStream_t st, int S, int n, const uint8_t *dgram, const int32_t *arfcn, int32_t *a_lf, int32_t *a_lk,
                                      int32_t *a_tot) {
  if (n <= 0) return hipSuccess;
  k_group_tx_arrive<<<dim3((S + kTxA - 1) / kTxA), dim3(1024), 0, st>>>(S, n, dgram, arfcn, tx_arrive_args(n, a_lf, a_lk, a_tot));
  return hipGetLastError();
}

hipError_t trx_launch_group_tx_ingest(hipStream_t st, const TrxGroupTx &x, int n, const uint8_t *dgram, const int32_t *a_lf, const int32_t *a_lk,
                                      const int32_t *a_tot, const float *gain_tab26, int ref_fn, int far) {
  if (n <= 0) return hipSuccess;
  if (x.qcap != kTxQ || x.npool > TRXQ_PK_IDS) return hipErrorInvalidValue;   // (the kernel's LDS copy of a queue, a packed entry's id field)
  TxGainTab gt;
  for (int q = 0; q < 26; q++) gt.v[q] = gain_tab26[q];
  k_group_tx<true, false><<<dim3((x.S + kTxI - 1) / kTxI), dim3(64 * kTxI), 0, st>>>(
      x, n, dgram, tx_arrive_args(n, (int32_t *)a_lf, (int32_t *)a_lk, (int32_t *)a_tot), gt, ref_fn, 0, 0, 0, nullptr, nullptr, nullptr, far);
  return hipGetLastError();
}

hipError_t trx_launch_group_tx_push(hipStream_t st, const TrxGroupTx &x, int fn0, int tn0, int n_slots, uint8_t *bits_out, float *gain_out,
                                    uint8_t *fq_out) {
  if (n_slots <= 0) return hipSuccess;
  if (x.qcap != kTxQ || x.npool > TRXQ_PK_IDS) return hipErrorInvalidValue;
  const int far = (long long)tn0 + n_slots >= 8LL * TRXQ_PK_WIN;    // the walk's own times must fit the packed form
  TxGainTab gt = {};
  TxArrive ar = {};
  k_group_tx<false, true><<<dim3((x.S + kTxI - 1) / kTxI), dim3(64 * kTxI), 0, st>>>(x, 0, nullptr, ar, gt, 0, fn0, tn0, n_slots, (uint32_t *)bits_out, gain_out,
                                                                                   fq_out, far);
  return hipGetLastError();
}

// both in one launch: the add call's ingest (its lists as left by trx_launch_group_tx_arrive) and then the push.  The packed entries are
// relative to fn0: the caller has seen to it that every datagram of the add call lies within TRXQ_PK_WIN frames of fn0 (else far).
hipError_t trx_launch_group_tx_both(hipStream_t st, const TrxGroupTx &x, int n, const uint8_t *dgram, const int32_t *a_lf, const int32_t *a_lk,
                                    const int32_t *a_tot, const float *gain_tab26, int far_add, int fn0, int tn0, int n_slots, uint8_t *bits_out,
                                    float *gain_out, uint8_t *fq_out) {
  if (n <= 0 || n_slots <= 0) return hipErrorInvalidValue;
  if (x.qcap != kTxQ || x.npool > TRXQ_PK_IDS) return hipErrorInvalidValue;
  const int far = far_add || (long long)tn0 + n_slots >= 8LL * TRXQ_PK_WIN;
  TxGainTab gt;
  for (int q = 0; q < 26; q++) gt.v[q] = gain_tab26[q];
  k_group_tx<true, true><<<dim3((x.S + kTxI - 1) / kTxI), dim3(64 * kTxI), 0, st>>>(
      x, n, dgram, tx_arrive_args(n, (int32_t *)a_lf, (int32_t *)a_lk, (int32_t *)a_tot), gt, fn0, fn0, tn0, n_slots, (uint32_t *)bits_out, gain_out, fq_out, far);
  return hipGetLastError();
}
