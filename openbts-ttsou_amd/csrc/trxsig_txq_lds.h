// trxsig_txq_lds.h -- internal: trxsig_txq.h's heap moves for a queue held in LDS (csrc/trxsig_grouptx.hip), written so that the
// host compiles them too: tests/test_txqueue_order.py runs them beside trxq_push / trxq_pop and std::priority_queue.
#pragma once
#include "trxsig_txq.h"

#define TRXQ_LDS_CAP 256                                    /* the queue's capacity: eight heap levels below the root */

#if defined(__HIPCC__)
typedef int2 TrxqEnt;                                       // x = fn, y = key (tn | id << 3)
#else
struct TrxqEnt { int32_t x, y; };
static inline int min(int a, int b) { return a < b ? a : b; }
#endif
TRXQ_HD TrxqEnt trxq_ent(int32_t fn, int32_t key) { TrxqEnt e; e.x = fn; e.y = key; return e; }

// ---- std::priority_queue's moves (trxsig_txq.h) on a queue in LDS, arranged so that a lane waits for LDS as seldom as the moves allow.
// The serial walk of a queue is a chain of dependent LDS round trips (~130 cycles each, a lane per ARFCN, nothing to hide them
// behind): trxq_push / trxq_pop as written make one or two per heap level.  The moves themselves leave room:
//   push  -- the path from the new leaf to the root is known before any comparison: every ancestor is fetched at once, the
//            comparisons run on registers (ONE round trip per push);
//   pop   -- __adjust_heap's hole goes down by whole levels: children and grandchildren are fetched together (two levels per round
//            trip; the first two levels come from registers, fetched after the previous pop), and the values moved up stay in
//            registers, so that __push_heap's climb back compares against them without reading anything.  Levels the last
//            element climbs back over end up holding what they held: only the levels above its final place are written.
// The element values, the comparisons (trxq_cmp) and the resulting array are those of trxq_push / trxq_pop, move for move
// (tests/test_gpu_trxgroup_tx.py holds the kernels against std::priority_queue itself, deep queues and ties included).
TRXQ_HD bool tx_gt(TrxqEnt a, TrxqEnt b) { return trxq_cmp(a.x, a.y, b.x, b.y); }
TRXQ_HD TrxqEnt tx_pick(bool c, TrxqEnt a, TrxqEnt b) { return trxq_ent(c ? a.x : b.x, c ? a.y : b.y); }

TRXQ_HD int tx_heap_push(TrxqEnt *row, int n, TrxqEnt v) {   // n = size before (< TRXQ_LDS_CAP); returns n + 1
  int idx[9];
  TrxqEnt a[9];
  idx[0] = n;
#pragma unroll
  for (int l = 1; l <= 8; l++) {                            // (TRXQ_LDS_CAP = 256: at most eight ancestors; past the root the root again)
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

// top: element 0 (in: as it is, out: as it is after the pop; undefined when the queue empties); c[0..5]: elements 1 .. 6 as they
// are (in and out; whatever lies past the queue's end is never looked at).  popped = the element handed out.  n = size before (> 0).
TRXQ_HD int tx_heap_pop(TrxqEnt *row, int n, TrxqEnt &top, TrxqEnt (&c)[6], TrxqEnt &popped) {
  popped = top;
  const int len = n - 1;
  if (len == 0) return 0;
  const TrxqEnt v = row[len];                                  // __pop_heap: value = *(last - 1), then __adjust_heap(first, 0, len, value)
  const int half = (len - 1) >> 1;                          // "while (secondChild < (len - 1) / 2)": both children exist
  const int lone = (len & 1) ? -1 : (len - 2) >> 1;         // the hole whose only child is element len - 1 (len even)
  int pos[9];
  TrxqEnt x[9];                                                // x[l]: the element moved up INTO level l - 1's hole, taken from pos[l]
  pos[0] = 0;
  int hole = 0, D = 0;
  bool go = true;
  TrxqEnt L = c[0], R = c[1], g0 = c[2], g1 = c[3], g2 = c[4], g3 = c[5];
#pragma unroll
  for (int s = 0; s < 4; s++) {                             // two levels a turn; 8 levels cover TRXQ_LDS_CAP = 256
    const int l1 = 2 * s + 1, l2 = 2 * s + 2;
    pos[l1] = pos[l2] = 0;
    x[l1] = x[l2] = v;
    if (go) {
      if (hole < half) {
        const bool lf = tx_gt(R, L);                        // "if (comp(first + secondChild, first + (secondChild - 1))) secondChild--"
        hole = 2 * hole + 2 - (int)lf;
        pos[l1] = hole; x[l1] = tx_pick(lf, L, R); D = l1;
        const TrxqEnt A = tx_pick(lf, g0, g2), B = tx_pick(lf, g1, g3);   // the chosen child's own children
        if (hole < half) {
          const bool lf2 = tx_gt(B, A);
          hole = 2 * hole + 2 - (int)lf2;
          pos[l2] = hole; x[l2] = tx_pick(lf2, A, B); D = l2;
        } else {
          go = false;
          if (hole == lone) { hole = 2 * hole + 1; pos[l2] = hole; x[l2] = A; D = l2; }
        }
      } else {
        go = false;
        if (hole == lone) { hole = 2 * hole + 1; pos[l1] = hole; x[l1] = L; D = l1; }
      }
      if (s < 3 && go) {                                    // the next two levels under the hole (clamped: past the end nothing is used)
        const int h2 = 2 * hole + 1, h4 = 4 * hole + 3;
        L = row[min(h2, TRXQ_LDS_CAP)]; R = row[min(h2 + 1, TRXQ_LDS_CAP)];
        g0 = row[min(h4, TRXQ_LDS_CAP)]; g1 = row[min(h4 + 1, TRXQ_LDS_CAP)]; g2 = row[min(h4 + 2, TRXQ_LDS_CAP)]; g3 = row[min(h4 + 3, TRXQ_LDS_CAP)];
      }
    }
  }
  // __push_heap(first, hole, 0, value): level j's parent holds x[j] now
  int j = D;
#pragma unroll
  for (int l = 8; l >= 1; l--)
    if (l == j && tx_gt(x[l], v)) j = l - 1;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    if (l < j) row[pos[l]] = x[l + 1];
    else if (l == j) row[pos[l]] = v;
  }
  top = j == 0 ? v : x[1];
#pragma unroll
  for (int i = 0; i < 6; i++) c[i] = row[1 + i];
  return len;
}

