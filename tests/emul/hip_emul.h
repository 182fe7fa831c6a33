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
// arrival-counting barrier over the `gsize`-lane group that contains the calling lane (64 = the wave)
void emul_barrier(int gsize);
#define EMUL_MAXT 256
extern uint64_t emul_xchg[EMUL_MAXT];
// wave-wide exchanges alternate between two buffers (emul_xchg2[parity of the lane's wave-collective count]): a lane can be at most one wave collective ahead of the
// slowest lane of its wave (it cannot pass the next one's barrier alone), so the buffer of exchange k is free again when exchange k + 2 writes it -- ONE barrier per
// shuffle / ballot instead of two (round 6: the register Newton step is ~900 broadcasts per factorisation)
extern uint64_t emul_xchg2[2][EMUL_MAXT];
extern unsigned emul_wave_calls[EMUL_MAXT];

static inline void __syncthreads() { emul_barrier((int)blockDim.x); }   // (a 64-thread workgroup is one wave; larger ones use the workgroup-wide barrier)

template <class T> static inline T emul_exchange(T v, int src, int gsize) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  if (gsize == 64) {
    uint64_t* buf = emul_xchg2[emul_wave_calls[threadIdx.x]++ & 1];
    buf[threadIdx.x] = raw;
    emul_barrier(64);
    T r; memcpy(&r, &buf[(threadIdx.x & ~63u) | (src & 63)], sizeof(T));
    return r;
  }
  emul_xchg[threadIdx.x] = raw;
  emul_barrier(gsize);
  T r; memcpy(&r, &emul_xchg[(threadIdx.x & ~63u) | (src & 63)], sizeof(T));   // (src: lane of the calling lane's wave)
  emul_barrier(gsize);
  return r;
}
// xor shuffles with a mask below 16 (8, 4, 2) stay inside a 16-lane row (8-lane half row, quad, lane pair): only that group has to be convergent
template <class T> static inline T __shfl_xor(T v, int mask) { return emul_exchange(v, (int)(threadIdx.x & 63) ^ mask, mask < 2 ? 2 : (mask < 4 ? 4 : (mask < 8 ? 8 : (mask < 16 ? 16 : 64)))); }
template <class T> static inline T __shfl(T v, int src) { return emul_exchange(v, src, 64); }
static inline unsigned long long __ballot(int pred) {
  uint64_t* buf = emul_xchg2[emul_wave_calls[threadIdx.x]++ & 1];
  buf[threadIdx.x] = pred ? 1 : 0;
  emul_barrier(64);
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (buf[(threadIdx.x & ~63u) + i]) r |= 1ull << i;
  return r;
}
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }

// launch: run `nblocks` workgroups of 64 fibers each; `fn(arg)` is the kernel body closure
typedef void (*emul_kernel_fn)(void* arg);
void emul_launch(int nblocks, size_t lds_bytes, emul_kernel_fn fn, void* arg);
void emul_launch_n(int nblocks, int nthreads, size_t lds_bytes, emul_kernel_fn fn, void* arg);   // workgroups of `nthreads` (a multiple of 64, <= EMUL_MAXT)
