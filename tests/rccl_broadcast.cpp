// The C-ABI's RCCL table broadcast (trxsig_tables_broadcast) from a plain C++ host, no torch: a one-rank communicator
// on GPU 0 (all a one-GPU box allows; with N ranks the call is the same), root uploads the host-built blob, broadcasts
// in place, creates a context from the received blob and checks that it holds the same tables as one built locally.
//   g++ -std=c++11 tests/rccl_broadcast.cpp -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lopenbts-ttsou_amd -ltrxsig
//       -L/opt/rocm/lib -lrccl -lamdhip64
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "trxsig.h"

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no GPU\n"); return 2; }
  hipSetDevice(0);
  ncclUniqueId id;
  ncclComm_t comm;
  if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "RCCL init failed\n"); return 3; }
  const size_t n = trxsig_tables_bytes(4);
  std::vector<unsigned char> h(n), back(n);
  if (trxsig_tables_build_host(4, h.data(), n) != TRXSIG_OK) return 4;
  void *d = nullptr;
  hipStream_t st;
  hipStreamCreate(&st);
  if (hipMalloc(&d, n) != hipSuccess || hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice) != hipSuccess) return 5;
  if (trxsig_tables_broadcast(comm, d, n, 0, st) != TRXSIG_OK) return 6;
  if (hipStreamSynchronize(st) != hipSuccess) return 7;
  trxsig_ctx *ctx = nullptr;
  if (trxsig_create_from_tables(&ctx, 0, d, n) != TRXSIG_OK) return 8;
  if (trxsig_tables_export(ctx, back.data(), n) != TRXSIG_OK || std::memcmp(back.data(), h.data(), n) != 0) return 9;
  // a corrupted blob must be refused
  unsigned char bad = h[5000] ^ 1;
  hipMemcpy((char *)d + 5000, &bad, 1, hipMemcpyHostToDevice);
  trxsig_ctx *ctx2 = nullptr;
  const int rc = trxsig_create_from_tables(&ctx2, 0, d, n);
  std::printf("broadcast ok, %zu bytes; corrupted blob -> %d\n", n, rc);
  trxsig_destroy(ctx);
  ncclCommDestroy(comm);
  hipFree(d);
  return rc == TRXSIG_OK ? 10 : 0;
}
