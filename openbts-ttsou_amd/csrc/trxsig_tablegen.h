// trxsig_tablegen.h -- host-side construction and validation of the TrxTables blob.
#pragma once
#include <stddef.h>
#include "trxsig_tables.h"

int trx_build_tables(TrxTables *T, int sps);          // 0 on success
uint32_t trx_tables_checksum(const TrxTables *T);
bool trx_tables_valid(const TrxTables *T);
const char *trx_training_sequence(int tsc);           // 26 characters '0'/'1', NULL if tsc is out of range
