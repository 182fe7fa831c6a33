// Measures how many single-wave workgroups with a given dynamic LDS size are resident per CU on this GPU
// (LDS allocation granularity is not documented in the guides): blocks spin for a fixed number of cycles; the
// elapsed time of a launch of 256 CUs x 24 blocks tells how many ran at once.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ldsprobe tools/lds_occupancy_probe.cpp && /tmp/ldsprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char lds_raw[];
__global__ void spin(long long cycles, int* sink) {
  long long t0 = __builtin_readcyclecounter();
  volatile unsigned char* p = lds_raw; p[threadIdx.x] = 1;
  while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && p[0] == 77) sink[0] = 1;
}
int main() {
  int sizes[] = {1024, 10240, 14080, 15360, 16384, 16640, 17920, 18176, 18204, 18432, 18736, 19200, 20480, 20736, 21760, 23040, 23360, 24320, 33728, 40960};
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int ncu = prop.multiProcessorCount;
  printf("CUs %d, clock %d kHz\n", ncu, prop.clockRate);
  const int per_cu = 48; const long long cycles = 2000000;   // s_memtime ticks (100 MHz constant clock on gfx9: 20 ms); calibrated below
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  // calibrate: one block per CU
  hipLaunchKernelGGL(spin, dim3(ncu), dim3(64), 1024, 0, 100000LL, (int*)nullptr); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(ncu), dim3(64), 1024, 0, 100000LL, (int*)nullptr); hipEventRecord(b); hipEventSynchronize(b);
  float t1; hipEventElapsedTime(&t1, a, b);
  printf("one block per CU spinning 100000 ticks: %.3f ms\n", t1);
  for (int s : sizes) {
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, s);
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(ncu * per_cu), dim3(64), s, 0, 100000LL, (int*)nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b);
    printf("LDS %6d B: %.3f ms -> ~%.2f rounds -> ~%.1f resident blocks per CU\n", s, t, t / t1, per_cu / (t / t1));
  }
  return 0;
}
