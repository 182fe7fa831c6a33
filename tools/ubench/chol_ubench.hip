// micro-benchmark (round 6): the register Newton factorisation of rg_kernel.h (rg_chol_inv_solve_n) in isolation.
// hipcc --offload-arch=gfx950 -O3 -o chol_ubench chol_ubench.hip && ./chol_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LANE ((int)threadIdx.x)
struct alignas(16) rgf4 { float x, y, z, w; };
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ float rg_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
template <int N, int MODE> __device__ __forceinline__ float elim(float* H, int hs4, float* x) {
  constexpr int NC = (N + 3) / 4;
  const int i = LANE, zr = i - (N + 1);
  const bool isrow = i <= N;
  float a[4 * NC];
  const rgf4* src = (const rgf4*)H + (isrow ? i : 0) * hs4;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const rgf4 v = src[c];
    a[4 * c + 0] = isrow ? v.x : (zr == 4 * c + 0 ? 1.f : 0.f);
    a[4 * c + 1] = isrow ? v.y : (zr == 4 * c + 1 ? 1.f : 0.f);
    a[4 * c + 2] = isrow ? v.z : (zr == 4 * c + 2 ? 1.f : 0.f);
    a[4 * c + 3] = isrow ? v.w : (zr == 4 * c + 3 ? 1.f : 0.f);
  }
  if (MODE == 1) {   // load/store only
  } else {
#pragma unroll
  for (int j = 0; j < N; j++) {
    float p = lane_bcast(a[j], j);
    const float inv = rg_rsqrt(fmaxf(p, 1e-30f));
    a[j] *= inv;
    const float naj = -a[j];
#pragma unroll
    for (int k = j + 1; k < N; k++) a[k] = __builtin_fmaf(naj, lane_bcast(a[j], k), a[k]);
  }
  }
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
  for (int k = 0; k < N; k += 4) {
    acc0 = __builtin_fmaf(a[k], lane_bcast(a[k], N), acc0);
    if (k + 1 < N) acc1 = __builtin_fmaf(a[k + 1], lane_bcast(a[k + 1], N), acc1);
    if (k + 2 < N) acc2 = __builtin_fmaf(a[k + 2], lane_bcast(a[k + 2], N), acc2);
    if (k + 3 < N) acc3 = __builtin_fmaf(a[k + 3], lane_bcast(a[k + 3], N), acc3);
  }
  float r = (acc0 + acc1) + (acc2 + acc3);
  if (zr >= 0 && zr < N) {
    x[zr] = r;
    rgf4* dst = (rgf4*)H + zr * hs4;
#pragma unroll
    for (int c = 0; c < NC; c++) { rgf4 o; o.x = a[4 * c]; o.y = a[4 * c + 1]; o.z = a[4 * c + 2]; o.w = a[4 * c + 3]; dst[c] = o; }
  }
  __syncthreads();
  return r;
}
extern __shared__ float lds[];
template <int MODE> __global__ void __launch_bounds__(64, 3) bench(const float* A, float* out, long long* cyc, int reps) {
  constexpr int N = 30, hs = 36;
  float* H = lds; float* x = lds + 31 * hs;
  float accum = 0;
  long long t0 = 0, total = 0;
  for (int r = 0; r < reps; r++) {
    for (int w = LANE; w < 31 * hs; w += 64) H[w] = A[w];
    __syncthreads();
    t0 = __builtin_readcyclecounter();
    accum += elim<N, MODE>(H, hs / 4, x);
    total += __builtin_readcyclecounter() - t0;
  }
  out[blockIdx.x * 64 + LANE] = accum + x[LANE % 30];
  if (LANE == 0) cyc[blockIdx.x] = total / reps;
}
int main() {
  const int N = 30, hs = 36;
  std::vector<float> A(31 * hs, 0.f);
  for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) A[i * hs + j] = (i == j) ? 10.f + i : 0.3f / (1 + i - j);
  for (int j = 0; j < N; j++) A[N * hs + j] = 1.f + 0.1f * j;
  float *dA, *dout; long long* dc;
  hipMalloc(&dA, A.size() * 4); hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  const int maxb = 256 * 12;
  hipMalloc(&dout, maxb * 64 * 4); hipMalloc(&dc, maxb * 8);
  for (int mode = 0; mode < 2; mode++) for (int perCU : {1, 4, 8, 12}) {
    int nb = 256 * perCU; size_t ldsb = 12656;
    for (int it = 0; it < 2; it++) {
      if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3(nb), dim3(64), ldsb, 0, dA, dout, dc, 200);
      else hipLaunchKernelGGL(bench<1>, dim3(nb), dim3(64), ldsb, 0, dA, dout, dc, 200);
      hipDeviceSynchronize();
    }
    std::vector<long long> c(nb); hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= nb;
    std::vector<float> o(64); hipMemcpy(o.data(), dout, 64 * 4, hipMemcpyDeviceToHost);
    printf("mode %d (%s) waves/CU %2d: %.0f cycles per call   (x[0] sum %g)\n", mode, mode == 0 ? "load+eliminate+dot+store" : "load+dot+store only", perCU, m, o[31]);
  }
  return 0;
}
