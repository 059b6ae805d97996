// dev_switches.h - every development-only switch of the kernels, in one place, compiled out of the product build by
// construction.
//
// The product library (million_amd/build.py, `make bindings`) defines NONE of these: MILLION_EXP is then the constant 0, every
// `#if MILLION_EXP & n` branch is dead text, and no environment variable changes which kernel form runs.  The A/B builds of
// tools/ab_build.py define MILLION_DEV_BUILD together with the switch they measure; a switch without MILLION_DEV_BUILD is a
// build error, so a product object can never carry a path the parity suite has not built.
//
//   MILLION_EXP (bit mask)
//     attn_mfma.hip   1  touch the code lines of rounds 2 and 3 early (one dword per line)        profiles/r04_launch_floor.txt
//                     2  the last-arriving workgroup merges alone (no helpers)                    profiles/r04_ab_merge.txt
//                    32  "the launch without arithmetic": every request, wait, barrier and the tail stay, a unit's bytes are
//                        xor-ed into a sink instead of gathered, multiplied and soft-maxed         profiles/r04_launch_floor.txt
//     prefill.hip     1  no exponentials, 2 no tile barrier, 4 no PV MFMAs, 8 no QK MFMAs, 16 no global -> LDS staging
//                                                                                                 profiles/r03_prefill.txt, r04_prefill.txt
//                   4 / 8 also in the pipelined kernel (no value / no score products);
//                   256  pipelined kernel: no exponentials, 512 no tile DMA behind the prologue, 1024 no tile wait/barrier  profiles/r05_prefill.txt
//                  2048  pipelined kernel: per-phase shader-clock sums of workgroup 0's waves, written into their own query rows (tools/prefill_prof.py)
//                 16384  pipelined kernel: only waves 0 - 3 compute (one computing wave per SIMD; its partner issues DMA and joins barriers)
//   MILLION_TILE_PROF      attn_tile.hip: per-tile shader-clock stamps (tools/tile_prof.py)
//   MILLION_DEV_M32_PACKED() run-time A/B (environment MILLION_M32_PACKED=1, dev builds only): M = 32 keeps the packed value form
//                          at up to 4 query heads per kv head too (the product build takes the d_m = 4 form there and the
//                          packed form above; both are parity-tested through those two ranges)
//
// MILLION_DEBUG_CHECK_IDS is NOT a development switch: it builds the shipped diagnostic variant libmillion_hip_dbgids.so
// (`make debug-ids`, include/million_hip.h), which the parity suite runs (tests/dbgids_child.py).
#pragma once

#ifndef MILLION_DEV_BUILD
#if defined(MILLION_EXP) || defined(MILLION_TILE_PROF)
#error "MILLION_EXP / MILLION_TILE_PROF are development switches: add -DMILLION_DEV_BUILD (tools/ab_build.py does); the product build defines neither"
#endif
#define MILLION_EXP 0
#define MILLION_DEV_M32_PACKED() 0
#else
#ifndef MILLION_EXP
#define MILLION_EXP 0
#endif
#include <stdlib.h>
#define MILLION_DEV_M32_PACKED() ([] { const char *e = getenv("MILLION_M32_PACKED"); return e && *e == '1' ? 1 : 0; }())
#endif
