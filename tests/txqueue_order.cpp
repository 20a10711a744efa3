// tests/txqueue_order.cpp -- holds the array form of the transmit priority queue (csrc/trxsig_txq.h: what the Transceiver group
// runs on the device, a lane per ARFCN) against the container the reference uses, std::priority_queue<T*, std::vector<T*>,
// PointerCompare> with the comparator *v1 > *v2 on GSM timestamps (CommonLibs/Interthread.h:432-463, radioInterface.h:58).
// stdin: "a fn tn id" = write, "p" = readNoBlock; stdout per p: "<id from std::priority_queue> <id from trxsig_txq.h>" (-1: empty).
#include <cstdio>
#include <queue>
#include <vector>

#include "trxsig_txq.h"

struct Item { int fn, tn, id; };
struct Later {                                              // PointerCompare<radioVector>: *v1 > *v2 -> GSM::Time::operator>
  bool operator()(const Item *a, const Item *b) const { return trxq_time_gt(a->fn, a->tn, b->fn, b->tn); }
};

int main() {
  std::priority_queue<Item *, std::vector<Item *>, Later> pq;
  std::vector<int32_t> fn(1 << 16), key(1 << 16);
  TrxqView q = {fn.data(), key.data(), 1};
  int n = 0;
  char op;
  while (std::scanf(" %c", &op) == 1) {
    if (op == 'a') {
      Item *it = new Item;
      if (std::scanf("%d %d %d", &it->fn, &it->tn, &it->id) != 3) return 2;
      pq.push(it);
      n = trxq_push(q, n, it->fn, it->tn | (it->id << 3));
    } else {
      int a = -1, b = -1;
      if (!pq.empty()) { Item *it = pq.top(); pq.pop(); a = it->id; delete it; }
      if (n > 0) { int32_t f, k; n = trxq_pop(q, n, &f, &k); b = k >> 3; }
      std::printf("%d %d\n", a, b);
    }
  }
  return 0;
}
