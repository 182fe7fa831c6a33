// hip_emul.h — TEST HARNESS ONLY.  Lets the HIP kernel source (robogym_amd/csrc/rg_kernel.h) be
// compiled by g++ and executed on the CPU: one ucontext fiber per lane, round-robin between
// barriers, so __syncthreads / shuffles / ballots behave as on a 64-lane wavefront.  It exists so
// that the kernel's arithmetic can be checked against the oracle without a GPU (the `-m "not gpu"`
// suite) and debugged with gdb/ASAN.  It is never built into, loaded by, or reachable from the
// product package (robogym_amd/); the product fails loudly without the gfx950 library and a GPU.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

struct emul_dim3 { unsigned x, y, z; };
extern emul_dim3 threadIdx, blockIdx, blockDim, gridDim;

void emul_yield();
void* emul_lds();
extern uint64_t emul_xchg[64];

static inline void __syncthreads() { emul_yield(); }

template <class T> static inline T emul_exchange(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  emul_xchg[threadIdx.x] = raw;
  emul_yield();
  T r; memcpy(&r, &emul_xchg[src & 63], sizeof(T));
  emul_yield();
  return r;
}
template <class T> static inline T __shfl_xor(T v, int mask) { return emul_exchange(v, (int)threadIdx.x ^ mask); }
template <class T> static inline T __shfl(T v, int src) { return emul_exchange(v, src); }
static inline unsigned long long __ballot(int pred) {
  emul_xchg[threadIdx.x] = pred ? 1 : 0;
  emul_yield();
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (emul_xchg[i]) r |= 1ull << i;
  emul_yield();
  return r;
}
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }

// launch: run `nblocks` workgroups of 64 fibers each; `fn(arg)` is the kernel body closure
typedef void (*emul_kernel_fn)(void* arg);
void emul_launch(int nblocks, size_t lds_bytes, emul_kernel_fn fn, void* arg);
