// hip_emul.cpp — fiber scheduler of the CPU emulation harness (see hip_emul.h).  TEST HARNESS ONLY.
#include "hip_emul.h"

#include <stdio.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <ucontext.h>
#include <setjmp.h>

emul_dim3 threadIdx, blockIdx, blockDim = {64, 1, 1}, gridDim;
uint64_t emul_xchg[EMUL_MAXT];
uint64_t emul_xchg2[2][EMUL_MAXT];
unsigned emul_wave_calls[EMUL_MAXT];

static const size_t kStack = 512 * 1024;
static ucontext_t g_sched, g_fiber[EMUL_MAXT];
static char* g_stacks = nullptr;
static int g_done[EMUL_MAXT];
static int g_nthreads = 64;
static int g_cur = 0;
static void* g_lds = nullptr;
static emul_kernel_fn g_fn;
static void* g_arg;

void* emul_lds() { return g_lds; }

// Fiber switches go through _setjmp / _longjmp (no signal-mask system call; swapcontext makes two per switch, which had become a third of the CPU suite's time once
// the register Newton step raised the number of wave collectives); ucontext only STARTS a fiber on its own stack.  (Built with -U_FORTIFY_SOURCE: the fortified
// longjmp refuses jumps between stacks.)
static jmp_buf g_sched_jb, g_fiber_jb[EMUL_MAXT];
static bool g_started[EMUL_MAXT];
void emul_yield() { if (_setjmp(g_fiber_jb[g_cur]) == 0) _longjmp(g_sched_jb, 1); }

// barrier state per group size (quads: 16 groups, 8-lane half rows: 8, 16-lane rows: 4, wave: 1, lane pairs: 32)
static int g_arrived[6][128], g_gen[6][128];
static int g_wsize[EMUL_MAXT];
static void* g_hist[EMUL_MAXT][16]; static long g_nwave[EMUL_MAXT];   // last wave-wide collectives of each lane + how many it has executed
static void* g_where[EMUL_MAXT];  // call site of the collective each lane waits in (deadlock report)
static bool g_restart;  // a wave barrier was released: resume the sweep at lane 0 (hardware executes a wave's lanes in lane order)
static int live_in_group(int gsize, int grp) { int n = 0; for (int l = grp * gsize; l < (grp + 1) * gsize; l++) n += !g_done[l]; return n; }
void emul_barrier(int gsize) {
  // group kinds: quad, half row, row, wave, lane pair, and the whole workgroup when it has more than one wave (gsize > 64)
  int k = gsize > 64 ? 5 : (gsize == 64 ? 3 : (gsize == 16 ? 2 : (gsize == 8 ? 1 : (gsize == 2 ? 4 : 0)))), gs = gsize > 64 ? g_nthreads : (gsize == 64 ? 64 : (gsize == 16 ? 16 : (gsize == 8 ? 8 : (gsize == 2 ? 2 : 4)))), grp = g_cur / gs;
  int gen = g_gen[k][grp];
  g_where[g_cur] = __builtin_return_address(0); g_wsize[g_cur] = gs;
  if (gs >= 64) { g_hist[g_cur][g_nwave[g_cur] & 15] = g_where[g_cur]; g_nwave[g_cur]++; }
  g_arrived[k][grp]++;
  if (g_arrived[k][grp] >= live_in_group(gs, grp)) { g_arrived[k][grp] = 0; g_gen[k][grp]++; g_restart = true; emul_yield(); return; }
  while (g_gen[k][grp] == gen) {
    // a lane of the group may have exited the kernel while we wait
    if (g_arrived[k][grp] >= live_in_group(gs, grp)) { g_arrived[k][grp] = 0; g_gen[k][grp]++; g_restart = true; emul_yield(); break; }
    emul_yield();
  }
}

static void fiber_main() {
  g_fn(g_arg);
  g_done[g_cur] = 1;
  _longjmp(g_sched_jb, 1);
}

void emul_launch(int nblocks, size_t lds_bytes, emul_kernel_fn fn, void* arg) { emul_launch_n(nblocks, 64, lds_bytes, fn, arg); }
void emul_launch_n(int nblocks, int nthreads, size_t lds_bytes, emul_kernel_fn fn, void* arg) {
  if (!g_stacks) g_stacks = (char*)malloc((size_t)EMUL_MAXT * kStack);
  g_nthreads = nthreads; blockDim.x = nthreads;
  g_fn = fn; g_arg = arg;
  gridDim.x = nblocks; gridDim.y = gridDim.z = 1;
  for (int b = 0; b < nblocks; b++) {
    g_lds = calloc(1, lds_bytes);
    // RG_EMUL_LDS_GARBAGE=1: LDS starts as on the hardware — whatever the previous workgroup left (here: NaN / huge-int patterns)
    { static const char* garb = getenv("RG_EMUL_LDS_GARBAGE"); if (garb && garb[0] == '1') { unsigned* u = (unsigned*)g_lds; for (size_t w = 0; w < lds_bytes / 4; w++) u[w] = (w & 1) ? 0x7fc00000u : 0xcdcdcdcdu; } }
    blockIdx.x = b; blockIdx.y = blockIdx.z = 0;
    for (int l = 0; l < nthreads; l++) {
      getcontext(&g_fiber[l]);
      g_fiber[l].uc_stack.ss_sp = g_stacks + l * kStack;
      g_fiber[l].uc_stack.ss_size = kStack;
      g_fiber[l].uc_link = &g_sched;
      makecontext(&g_fiber[l], fiber_main, 0);
      g_done[l] = 0; g_started[l] = false;
    }
    for (int k = 0; k < 6; k++) for (int g = 0; g < 128; g++) g_arrived[k][g] = 0;
    for (int l = 0; l < nthreads; l++) { g_nwave[l] = 0; emul_wave_calls[l] = 0; }
    volatile int alive = nthreads;
    volatile long idle_sweeps = 0;
    while (alive > 0) {
      alive = 0;
      // a sweep over all lanes that releases no barrier makes no progress: lanes wait in DIFFERENT collectives (control flow
      // around a wave collective that is not uniform).  Report where instead of spinning.
      if (++idle_sweeps > 100000) {
        fprintf(stderr, "hip_emul: deadlock in block %d; lanes wait at (addr2line -e librgstep_emul.so <offset>):\n", b);
        for (int l = 0; l < nthreads; l++) {
          Dl_info info; size_t off = (size_t)g_where[l];
          if (dladdr(g_where[l], &info)) off -= (size_t)info.dli_fbase;
          fprintf(stderr, "  lane %2d %s 0x%zx (group of %d)\n", l, g_done[l] ? "exited " : "waiting", off, g_wsize[l]);
        }
        for (int l = 0; l < nthreads; l++) {
          if (l && g_nwave[l] == g_nwave[l - 1]) continue;
          fprintf(stderr, "  lane %d has executed %ld wave-wide collectives, the last at:", l, g_nwave[l]);
          for (long q = g_nwave[l] - 1; q >= 0 && q >= g_nwave[l] - 12; q--) { Dl_info info; size_t off = (size_t)g_hist[l][q & 15]; if (dladdr(g_hist[l][q & 15], &info)) off -= (size_t)info.dli_fbase; fprintf(stderr, " 0x%zx", off); }
          fprintf(stderr, "\n");
        }
        abort();
      }
      for (volatile int l = 0; l < nthreads; l++) {
        if (g_done[l]) continue;
        g_cur = l; threadIdx.x = l; threadIdx.y = threadIdx.z = 0;
        if (_setjmp(g_sched_jb) == 0) {
          if (!g_started[l]) { g_started[l] = true; setcontext(&g_fiber[l]); }
          else _longjmp(g_fiber_jb[l], 1);
        }
        if (!g_done[l]) alive++;
        if (g_restart) { g_restart = false; alive = nthreads; l = -1; idle_sweeps = 0; }
      }
    }
    free(g_lds);
  }
}
