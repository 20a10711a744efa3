#include <cstdio>
#include <initializer_list>
#include "trxsig_tablegen.h"
int main() {
  static TrxTables T;
  for (int sps : {1, 2, 4}) {
    if (trx_build_tables(&T, sps) != 0) { std::printf("build failed %d\n", sps); return 1; }
    if (!trx_tables_valid(&T)) { std::printf("invalid %d\n", sps); return 2; }
    std::printf("sps %d ok checksum %08x\n", sps, trx_tables_checksum(&T));
  }
  for (int t = -1; t < 9; t++) (void)trx_training_sequence(t);
  return 0;
}
