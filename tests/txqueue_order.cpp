// tests/txqueue_order.cpp -- holds the array form of the transmit priority queue (csrc/trxsig_txq.h: what the Transceiver group
// runs on the device, a lane per ARFCN) against the container the reference uses, std::priority_queue<T*, std::vector<T*>,
// PointerCompare> with the comparator *v1 > *v2 on GSM timestamps (CommonLibs/Interthread.h:432-463, radioInterface.h:58).
// The third column is the same queue in the form the kernels keep it in LDS (csrc/trxsig_txq_lds.h: tx_heap_push / tx_heap_pop, ancestors
// fetched at once, two levels a turn): besides the popped ids, its ARRAY must equal trxsig_txq.h's after every operation (exit code 3).
// stdin: "a fn tn id" = write, "p" = readNoBlock; stdout per p: "<id from std::priority_queue> <id from trxsig_txq.h> <id from the LDS form>"
// (-1: empty; the LDS form's id is the low 11 bits).  Writes beyond TRXQ_LDS_CAP - 1 entries are not sent to the LDS form's queue (the kernels drop them before the push).
#include <cstdio>
#include <queue>
#include <vector>

#include "trxsig_txq.h"
#include "trxsig_txq_lds.h"

struct Item { int fn, tn, id; };
struct Later {                                              // PointerCompare<radioVector>: *v1 > *v2 -> GSM::Time::operator>
  bool operator()(const Item *a, const Item *b) const { return trxq_time_gt(a->fn, a->tn, b->fn, b->tn); }
};

int main() {
  std::priority_queue<Item *, std::vector<Item *>, Later> pq;
  std::vector<int32_t> fn(1 << 16), key(1 << 16);
  TrxqView q = {fn.data(), key.data(), 1};
  int n = 0;
  std::vector<TrxqPk> row(TRXQ_LDS_CAP + 1);                // the packed form: one word an entry, relative to the first write's frame
  TrxqPk top = 0, c[6] = {0, 0, 0, 0, 0, 0}, last = 0;
  int ref = -1;
  auto is = [&](TrxqPk e, int i) {
    return trxq_pk_fn(e, ref) == fn[i] && trxq_pk_tn(e) == (key[i] & 7) && trxq_pk_id(e) == ((key[i] >> 3) & (TRXQ_PK_IDS - 1));
  };
  auto same = [&]() {
    for (int i = 0; i < n; i++) if (!is(row[i], i)) return false;
    if (n > 0 && (!is(top, 0) || !is(last, n - 1))) return false;
    for (int i = 0; i < 6 && 1 + i < n; i++) if (!is(c[i], 1 + i)) return false;
    return true;
  };
  char op;
  while (std::scanf(" %c", &op) == 1) {
    if (op == 'a') {
      Item *it = new Item;
      if (std::scanf("%d %d %d", &it->fn, &it->tn, &it->id) != 3) return 2;
      pq.push(it);
      if (n >= TRXQ_LDS_CAP) return 4;
      if (ref < 0) ref = it->fn;
      if (!trxq_pk_ok(it->fn, ref)) return 5;
      const int m = tx_heap_push(row.data(), n, trxq_pk(it->fn, it->tn, it->id & (TRXQ_PK_IDS - 1), ref));
      n = trxq_push(q, n, it->fn, it->tn | (it->id << 3));
      top = row[0];                                         // (the kernels that push do not pop: the walk loads these when it starts)
      for (int i = 0; i < 6; i++) c[i] = row[1 + i];
      last = row[n - 1];
      if (m != n || !same()) return 3;
    } else {
      int a = -1, b = -1, l = -1;
      if (!pq.empty()) { Item *it = pq.top(); pq.pop(); a = it->id; delete it; }
      if (n > 0) {
        TrxqPk e;
        const int m = tx_heap_pop(row.data(), n, top, c, last, e);
        l = trxq_pk_id(e);
        int32_t f, k; n = trxq_pop(q, n, &f, &k); b = k >> 3;
        if (m != n || trxq_pk_fn(e, ref) != f || trxq_pk_tn(e) != (k & 7) || l != (b & (TRXQ_PK_IDS - 1)) || !same()) return 3;
      }
      std::printf("%d %d %d\n", a, b, l);
    }
  }
  return 0;
}
