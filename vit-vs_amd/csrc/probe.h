// Probe builds (-DVITVS_PROBE: tools/gemm_probe.cpp, tools/big_ops fixed / probe / attn, tools/measure_round.sh) stamp the phases of
// the GEMM and attention kernels with the cycle counter and hand the stamps out through a device pointer.  In the product build
// every one of these statements compiles to nothing, so the kernels read as what ships: the hand-counted waits stay in view.
//
//   VITVS_IF_PROBE(statements)      the statements in probe builds, nothing otherwise
//   VITVS_STAMP(var)                var = cycle counter, fenced against the scheduler on both sides (probe builds only)
//   VITVS_PROBE_OR_NULL(ptr)        ptr in probe builds, nullptr otherwise (a stamp array handed to a shared main loop)
// The device pointers the stamps go to, and the vitvs_debug_set_* entry points that set them, are the only probe code left at
// file scope of gemm.hip / gemm_big.hip / attention.hip (one #ifdef each).
#pragma once

#ifdef VITVS_PROBE
#define VITVS_IF_PROBE(...) __VA_ARGS__
#define VITVS_PROBE_OR_NULL(ptr) (ptr)
#define VITVS_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VITVS_IF_PROBE(...)
#define VITVS_PROBE_OR_NULL(ptr) nullptr
#define VITVS_STAMP(var) do { } while (0)
#endif
