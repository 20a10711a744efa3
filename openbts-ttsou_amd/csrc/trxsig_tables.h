// trxsig_tables.h -- the constant-table blob shared by host and device code.
//
// One POD struct, identical layout on host and device, uploaded once per context (or received
// by RCCL broadcast on non-root ranks).  It carries what the reference keeps in
// process globals (Transceiver/sigProcLib.cpp:39-59) plus two derived tables the kernels use:
//   mid_ctap  conj() of the 16 non-zero taps of each midamble sequence (the unit-pulse midamble
//             is non-zero only every sps-th sample, sigProcLib.cpp:794-797)
//   sinc_grid sinc(M_PI_F*(j-10-f/512)) for f=0..511, j=0..20: every argument interpolatePoint
//             (sigProcLib.cpp:651) and delayVector (:588) can form once TOA lies on peakDetect's
//             1/512-sample grid (:687-700)
#pragma once
#include <stdint.h>

#define TRX_MAXSPS 4
#define TRX_TABLESIZE 1024
#define TRX_MAGIC 0x54525853u /* "TRXS" */
#define TRX_BLOB_VERSION 2u
#define TRX_SINC_ROW 32   /* one 128-byte line per row */

struct trx_c32 { float r, i; };

struct TrxTables {
  uint32_t magic, version, sps, bytes;
  uint32_t checksum, pad0[3];
  float cosT[TRX_TABLESIZE + 4];            // 1025 used; [1025] = 0 guard (arg == 1.0 reads it times 0)
  float sinT[TRX_TABLESIZE + 4];
  trx_c32 rot[157 * TRX_MAXSPS];            // GMSKRotation
  trx_c32 rev[157 * TRX_MAXSPS];            // GMSKReverseRotation
  float pulse[2 * TRX_MAXSPS + 4];          // generateGSMPulse(2,sps): 2*sps+1 real taps
  trx_c32 mid[8][16 * TRX_MAXSPS];          // gMidambles[t]->sequence
  float mid_toa[8];                         // Transceiver/ variant (TOA - 5*sps)
  trx_c32 mid_gain[8];
  trx_c32 rach[41 * TRX_MAXSPS];            // gRACHSequence->sequence
  float rach_toa, pad1;
  trx_c32 rach_gain;
  trx_c32 mid_ctap[8][16];                  // conj(mid[t][sps*k]), k = 0..15
  float pad2[16];                           // puts sinc_grid on a 128-byte boundary of the (256-byte aligned) blob
  float sinc_grid[512][TRX_SINC_ROW];       // [f][j], j < 21, rest 0; a row is exactly one cache line
};
static_assert(__builtin_offsetof(TrxTables, sinc_grid) % 128 == 0, "sinc_grid rows are cache lines");
