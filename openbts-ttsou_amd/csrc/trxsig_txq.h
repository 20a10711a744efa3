// trxsig_txq.h -- internal: the transmit priority queue of class Transceiver (mTransmitPriorityQueue, Transceiver.h:72) as
// plain arrays, for host and device code alike.
//
// The reference's queue is VectorQueue = InterthreadPriorityQueue<radioVector> (Transceiver/radioInterface.h:64-72,
// CommonLibs/Interthread.h:432-528): a std::priority_queue<radioVector*, std::vector<radioVector*>, PointerCompare> whose
// comparator is `*v1 > *v2`, i.e. radioVector::operator> = GSM::Time::operator> on the bursts' timestamps
// (radioInterface.h:58, GSM/GSMCommon.h:431-435, FNCompare / FNDelta GSMCommon.cpp:161-176).  Which of two bursts with EQUAL
// timestamps leaves first is decided by the heap's shape, so the shape is reproduced: push and pop below are libstdc++'s
// std::push_heap / std::pop_heap (bits/stl_heap.h: __push_heap, __adjust_heap, __pop_heap) step for step, on (fn, tn, id)
// triples instead of pointers.  tests/test_txqueue_order.py holds it against std::priority_queue itself (and against the
// Python model's heapq, which makes the same moves).
//
// Storage: element i of a queue lives at fn[i * stride], key[i * stride] -- stride 1 on the host, the number of ARFCNs on the
// device (a lane per ARFCN walks its own queue, neighbouring lanes touch neighbouring words).  key = tn | id << 3.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TRXQ_HD __host__ __device__ __forceinline__
#else
#define TRXQ_HD inline
#endif

#define TRXQ_HYPERFRAME (2048 * 26 * 51)                    /* gHyperframe, GSM/GSMCommon.h:306 */

// GSM::FNDelta (GSMCommon.cpp:161-168)
TRXQ_HD int32_t trxq_fn_delta(int32_t v1, int32_t v2) {
  const int32_t half = TRXQ_HYPERFRAME / 2;
  int32_t delta = v1 - v2;
  if (delta >= half) delta -= TRXQ_HYPERFRAME;
  else if (delta < -half) delta += TRXQ_HYPERFRAME;
  return delta;
}
// GSM::Time::operator> / operator< / operator== (GSMCommon.h:425-449)
TRXQ_HD bool trxq_time_gt(int32_t fn1, int tn1, int32_t fn2, int tn2) { return fn1 == fn2 ? tn1 > tn2 : trxq_fn_delta(fn1, fn2) > 0; }
TRXQ_HD bool trxq_time_lt(int32_t fn1, int tn1, int32_t fn2, int tn2) { return fn1 == fn2 ? tn1 < tn2 : trxq_fn_delta(fn1, fn2) < 0; }

struct TrxqView {
  int32_t *fn;                                              // element i at fn[i * stride]
  int32_t *key;                                             // tn | id << 3
  int stride;
  TRXQ_HD int32_t f(int i) const { return fn[i * stride]; }
  TRXQ_HD int32_t k(int i) const { return key[i * stride]; }
  TRXQ_HD void set(int i, int32_t vf, int32_t vk) const { fn[i * stride] = vf; key[i * stride] = vk; }
};
// (Round 5 also tried the queue in ONE WAVE's registers -- element i in lane i & 63 of register i >> 6, v_readlane reads, compare-and-select
//  writes, a wave per ARFCN: rocprofv3 put k_group_tx_push at 247 us against 126 with the queues in LDS -- every access a readfirstlane,
//  a four-way register select and a lane read, ~10 dependent scalar-vector hops where LDS needs one round trip; not kept.  The heap
//  moves below are templates over the queue's storage since then.)
TRXQ_HD bool trxq_cmp(int32_t fn1, int32_t key1, int32_t fn2, int32_t key2) {   // PointerCompare: *v1 > *v2
  return trxq_time_gt(fn1, key1 & 7, fn2, key2 & 7);
}
// std::__push_heap(first, holeIndex, topIndex, value, comp)
template <class Q>
TRXQ_HD void trxq_sift_up(Q &q, int hole, int top, int32_t vfn, int32_t vkey) {
  int parent = (hole - 1) / 2;
  while (hole > top && trxq_cmp(q.f(parent), q.k(parent), vfn, vkey)) {
    q.set(hole, q.f(parent), q.k(parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  q.set(hole, vfn, vkey);
}
// priority_queue::push: c.push_back(value); std::push_heap(c.begin(), c.end()).  n = size before; returns the new size
template <class Q>
TRXQ_HD int trxq_push(Q &q, int n, int32_t vfn, int32_t vkey) {
  trxq_sift_up(q, n, 0, vfn, vkey);
  return n + 1;
}
// priority_queue::pop: std::pop_heap(c.begin(), c.end()); c.pop_back().  The top (element 0) is handed out through
// *tfn / *tkey; n = size before (> 0); returns the new size
template <class Q>
TRXQ_HD int trxq_pop(Q &q, int n, int32_t *tfn, int32_t *tkey) {
  *tfn = q.f(0);
  *tkey = q.k(0);
  if (n > 1) {
    // __pop_heap(first, last - 1, last - 1): value = *(last - 1); __adjust_heap(first, 0, len = n - 1, value)
    const int len = n - 1;
    const int32_t vfn = q.f(len), vkey = q.k(len);
    int hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      if (trxq_cmp(q.f(second), q.k(second), q.f(second - 1), q.k(second - 1))) second--;
      q.set(hole, q.f(second), q.k(second));
      hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      q.set(hole, q.f(second - 1), q.k(second - 1));
      hole = second - 1;
    }
    trxq_sift_up(q, hole, 0, vfn, vkey);
  }
  return n - 1;
}
