// trxsig_trxstate.h -- internal: the control-plane state of one `Transceiver` (Transceiver/Transceiver.h:83-116, set through
// driveControl, Transceiver.cpp:439-580) and the slot schedule that hangs off it (expectedCorrType :207-269, setModulus
// :183-204).  Shared by the one-ARFCN object (trxsig_transceiver.cpp) and the group of S of them (trxsig_trxgroup.cpp).
#pragma once
#include "trxsig_transceiver.h"

struct TrxControl {
  bool on = false;
  double txFreq = 0.0, rxFreq = 0.0;
  int power = -10;
  unsigned tsc = 0;
  int maxDelay = 0;                                         // mMaxExpectedDelay (Transceiver52M/Transceiver.cpp:62, SETMAXDELAY :476-486; stored, the TSC leg is chosen by trxsig_*_set_tsc_leg)
  int chanType[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // TRXSIG_CHAN_NONE
  int fillerModulus[8] = {26, 26, 26, 26, 26, 26, 26, 26};
  unsigned epoch = 0;                                       // bumped whenever chanType / tsc change (the group re-derives its tables)

  // driveControl's command switch: `buffer` = the NUL-terminated datagram (< 100 bytes), `response` = room for 100.
  // Returns 1 with the response filled, 0 where the reference sends nothing ("bogus message", SETSLOT out of range).
  int command(const char *buffer, char *response);
  int expectedCorrType(int tn, int fn) const;               // TRXSIG_CORR_*
  static int corrType(int chanType, int fn);
  void setModulus(int ts);
};
