// Row lengths P = m+1 the conditioning-set kernel is compiled for.  A plan whose
// m+1 is not in the list is padded (identity rows) to the next larger entry.
// 11/21/31/61 are BASELINE.json's configs (m = 10/20/30/60); the rest bound the padding waste.
#pragma once
#ifndef GPV_P_LIST          // (a tuning build may pass a shorter list: python -m gpvecchia_amd.build --tag _x --plist 31)
#define GPV_P_LIST(X) X(4) X(8) X(11) X(16) X(21) X(26) X(31) X(32) X(41) X(51) X(61) X(64)
#endif
