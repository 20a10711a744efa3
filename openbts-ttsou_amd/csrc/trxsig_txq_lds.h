// trxsig_txq_lds.h -- internal: trxsig_txq.h's heap moves for a queue held in LDS (csrc/trxsig_grouptx.hip), written so that the
// host compiles them too: tests/txqueue_order.cpp runs them beside trxq_push / trxq_pop and std::priority_queue.
//
// The serial walk of a queue -- a lane per ARFCN -- is bound by the instructions a wave issues and by dependent LDS round trips
// (profiles/r05_group_tx_probe.txt); trxq_push / trxq_pop as written cost ~15 instructions of comparison and two to four reads a
// level.  Two things are done about it here, neither changes a move:
//
// (1) AN ENTRY IS ONE WORD.  While every burst of a queue lies within +-2^17 frames (ten minutes) of a reference frame `ref`, the
//     wrap-aware comparison GSM::Time::operator> (FNDelta, GSMCommon.cpp:161-176) IS the integer comparison of
//     FNDelta(fn, ref) * 8 + tn: both offsets lie inside a quarter hyperframe, so FNDelta(fn1, fn2) = d1 - d2 without a wrap and
//     fn1 == fn2 exactly when d1 == d2.  The word is [31:14] d (signed) | [13:11] tn | [10:0] payload id; trxq_cmp becomes a shift
//     and a compare.  A queue holding anything further away is walked by trxq_push / trxq_pop on the arrays in memory instead (the
//     kernels' slow path: same results, the round-4 speed).
// (2) A LANE WAITS FOR LDS AS SELDOM AS THE MOVES ALLOW.
//     push -- the path from the new leaf to the root is known before any comparison: every ancestor is fetched at once, the
//             comparisons run on registers (one round trip a push);
//     pop  -- __adjust_heap's hole goes down two levels a round trip (children and grandchildren fetched together; the first two
//             levels and the queue's last element come from registers, fetched after the previous pop), and the element last moved
//             up is still in a register when __push_heap's climb back starts: its first comparison -- usually the only one --
//             reads nothing.
// The array after every operation is trxq_push's / trxq_pop's array (unpacked), which is std::priority_queue's.
#pragma once
#include "trxsig_txq.h"

#define TRXQ_LDS_CAP 256                                    /* the queue's capacity: eight heap levels below the root */
#define TRXQ_PK_WIN (1 << 17)                               /* frames either side of the reference a packed entry can say */
#define TRXQ_PK_IDS (1 << 11)                               /* payload ids a packed entry can say */

typedef int32_t TrxqPk;

#if !defined(__HIPCC__)
static inline int min(int a, int b) { return a < b ? a : b; }
#endif

TRXQ_HD bool trxq_pk_ok(int32_t fn, int32_t ref) {
  const int32_t d = trxq_fn_delta(fn, ref);
  return d >= -TRXQ_PK_WIN && d < TRXQ_PK_WIN;
}
TRXQ_HD TrxqPk trxq_pk(int32_t fn, int tn, int id, int32_t ref) {
  return (TrxqPk)(((uint32_t)trxq_fn_delta(fn, ref) << 14) | ((uint32_t)tn << 11) | (uint32_t)id);
}
TRXQ_HD int32_t trxq_pk_fn(TrxqPk e, int32_t ref) {
  int32_t fn = ref + (e >> 14);
  fn -= fn >= TRXQ_HYPERFRAME ? TRXQ_HYPERFRAME : 0;
  fn += fn < 0 ? TRXQ_HYPERFRAME : 0;
  return fn;
}
TRXQ_HD int trxq_pk_tn(TrxqPk e) { return (e >> 11) & 7; }
TRXQ_HD int trxq_pk_id(TrxqPk e) { return e & (TRXQ_PK_IDS - 1); }
TRXQ_HD int32_t trxq_pk_time(TrxqPk e) { return e >> 11; }    // FNDelta(fn, ref) * 8 + tn: what the queue orders by
TRXQ_HD bool tx_gt(TrxqPk a, TrxqPk b) { return (a >> 11) > (b >> 11); }   // PointerCompare: *v1 > *v2

// priority_queue::push.  n = size before (< TRXQ_LDS_CAP); returns n + 1.  row: TRXQ_LDS_CAP + 1 words.
TRXQ_HD int tx_heap_push(TrxqPk *row, int n, TrxqPk v) {
  int idx[9];
  TrxqPk a[9];
  idx[0] = n;
#pragma unroll
  for (int l = 1; l <= 8; l++) {                            // (at most eight ancestors; past the root the root again)
    idx[l] = idx[l - 1] > 0 ? (idx[l - 1] - 1) >> 1 : 0;
    a[l] = row[idx[l]];
  }
  int hole = n;
#pragma unroll
  for (int l = 1; l <= 8; l++) {                            // std::__push_heap: while (hole > top && comp(first[parent], value))
    if (hole == 0 || !tx_gt(a[l], v)) break;
    row[hole] = a[l];
    hole = idx[l];
  }
  row[hole] = v;
  return n + 1;
}

// priority_queue::pop.  top: element 0 (in: as it is; out: as it is after the pop, undefined when the queue empties); c[0..5]: elements
// 1 .. 6 likewise (whatever lies past the queue's end is never looked at); last: element n - 1 likewise.  popped = the element
// handed out.  n = size before (> 0).
TRXQ_HD int tx_heap_pop(TrxqPk *row, int n, TrxqPk &top, TrxqPk (&c)[6], TrxqPk &last, TrxqPk &popped) {
  popped = top;
  const int len = n - 1;
  if (len == 0) return 0;
  const TrxqPk v = last;                                    // __pop_heap: value = *(last - 1), then __adjust_heap(first, 0, len, value)
  const int half = (len - 1) >> 1;                          // "while (secondChild < (len - 1) / 2)": both children exist
  const int lone = (len & 1) ? -1 : (len - 2) >> 1;         // the hole whose only child is element len - 1 (len even)
  int hole = 0;
  TrxqPk xl = v, w0 = v;                                    // the element last moved up (it sits in the hole's parent); what element 0 became
  TrxqPk L = c[0], R = c[1], g0 = c[2], g1 = c[3], g2 = c[4], g3 = c[5];
  for (;;) {                                                // two levels a turn
    if (hole >= half) {
      if (hole == lone) { row[hole] = L; xl = L; w0 = hole == 0 ? L : w0; hole = 2 * hole + 1; }
      break;
    }
    const bool lf = tx_gt(R, L);                            // "if (comp(first + secondChild, first + (secondChild - 1))) secondChild--"
    xl = lf ? L : R;
    row[hole] = xl;
    w0 = hole == 0 ? xl : w0;
    hole = 2 * hole + 2 - (int)lf;
    const TrxqPk A = lf ? g0 : g2, B = lf ? g1 : g3;        // the chosen child's own children
    if (hole >= half) {
      if (hole == lone) { row[hole] = A; xl = A; hole = 2 * hole + 1; }
      break;
    }
    const bool lf2 = tx_gt(B, A);
    xl = lf2 ? A : B;
    row[hole] = xl;
    hole = 2 * hole + 2 - (int)lf2;
    const int h2 = 2 * hole + 1, h4 = 4 * hole + 3;         // the next two levels under the hole (clamped: past the end nothing is used)
    L = row[min(h2, TRXQ_LDS_CAP)]; R = row[min(h2 + 1, TRXQ_LDS_CAP)];
    g0 = row[min(h4, TRXQ_LDS_CAP)]; g1 = row[min(h4 + 1, TRXQ_LDS_CAP)];
    g2 = row[min(h4 + 2, TRXQ_LDS_CAP)]; g3 = row[min(h4 + 3, TRXQ_LDS_CAP)];
  }
  while (hole > 0 && tx_gt(xl, v)) {                        // __push_heap(first, hole, 0, value): xl is what the hole's parent holds
    row[hole] = xl;
    hole = (hole - 1) >> 1;
    if (hole > 0) xl = row[(hole - 1) >> 1];
  }
  row[hole] = v;
  top = hole == 0 ? v : w0;
#pragma unroll
  for (int i = 0; i < 6; i++) c[i] = row[1 + i];
  last = row[len - 1];
  return len;
}
