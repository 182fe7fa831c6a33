// micro-benchmark + correctness check (round 6): x = inv(H) g for a 30 x 30 SPD H by a right-looking elimination on the MFMA pipe.
// Two 32 x 32 f32 accumulator tiles: S = [H g] (symmetric storage) and T' = transpose of the rows that start as the identity; every pair of
// columns is one rank-2 update v_mfma_f32_32x32x2_f32 per tile.  See rg_kernel.h rg_chol_mfma.
// hipcc --offload-arch=gfx950 -O3 -o mfma_chol_ubench mfma_chol_ubench.hip && ./mfma_chol_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define LANE ((int)threadIdx.x)
struct alignas(16) rgf4 { float x, y, z, w; };
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ float rg_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
// (lo half of a, lo half of b) if !H else (hi half of a, hi half of b)
template <int H> __device__ __forceinline__ float halves(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, H ? r[1] : r[0]);
}
template <int N> __device__ __forceinline__ void chol_mfma(float* Hm, int hs, float* x) {
  const int l = LANE, col = l & 31, hi = l >> 5, hs4 = hs >> 2;
  const rgf4* H4 = (const rgf4*)Hm;
  f16v S, T;
  // ---- load
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const int i0 = 8 * g + 4 * hi;
    const int cc = col < N + 1 ? col : N;                 // (column 31: clamped, zeroed below)
    const rgf4 rowf = H4[cc * hs4 + 2 * g + hi];          // H[c][i0..i0+3]: valid where i <= c
    float v[4] = {rowf.x, rowf.y, rowf.z, rowf.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = i0 + q, ic = i < N + 1 ? i : N;
      const float colf = Hm[ic * hs + cc];                // H[i][c]: valid where c <= i
      float val = i < cc ? v[q] : colf;
      if (i > N || col > N) val = 0.f;
      S[4 * g + q] = val;
      T[4 * g + q] = (i == col && col < N) ? 1.f : 0.f;
    }
  }
  const int cm = col - hi;
  // ---- eliminate, two columns per step
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    constexpr int dummy = 0; (void)dummy;
    const int rj = 4 * (j >> 3) + (j & 3), hj = (j >> 2) & 1, lj = 32 * hj + j;
    const float s0 = S[rj], s1 = S[rj + 1], t0 = T[rj], t1 = T[rj + 1];
    const float p = lane_bcast(s0, lj);
    const float inv = rg_rsqrt(fmaxf(p, 1e-30f));
    const float w0 = s0 * inv, z0 = t0 * inv;
    const float mlt = lane_bcast(w0, lj + 1);
    const float s1c = __builtin_fmaf(-mlt, w0, s1), t1c = __builtin_fmaf(-mlt, z0, t1);
    const float p2 = lane_bcast(s1c, lj + 1);
    const float inv2 = rg_rsqrt(fmaxf(p2, 1e-30f));
    const float w1 = s1c * inv2, z1 = t1c * inv2;
    const bool mine = hi == hj;
    T[rj] = mine ? z0 : t0; T[rj + 1] = mine ? z1 : t1;
    float WW, ZZ;
    if (hj == 0) { WW = halves<0>(w0, w1); ZZ = halves<0>(z0, z1); } else { WW = halves<1>(w0, w1); ZZ = halves<1>(z0, z1); }
    const float WWn = col > j + 1 ? -WW : 0.f;   // rows <= j + 1 are final (row j + 1 took its correction on the VALU)
    S = __builtin_amdgcn_mfma_f32_32x32x2f32(WWn, WW, S, 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x2f32(WWn, ZZ, T, 0, 0, 0);
  }
  // ---- x_i = -T'[N][i]  (row N = 30: g 3, h 1, q 2 -> register 14, upper half)
  constexpr int rN = 4 * (N >> 3) + (N & 3), hN = (N >> 2) & 1;
  if (hi == hN && col < N) x[col] = -T[rN];
  // ---- W[m][k] = T'[k][m] for the factor-reusing iterations
  if (col < N) {
    rgf4* dst = (rgf4*)Hm + col * hs4;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int k0 = 8 * g + 4 * hi;
      rgf4 o; o.x = k0 < N ? T[4 * g] : 0.f; o.y = k0 + 1 < N ? T[4 * g + 1] : 0.f; o.z = k0 + 2 < N ? T[4 * g + 2] : 0.f; o.w = k0 + 3 < N ? T[4 * g + 3] : 0.f;
      dst[2 * g + hi] = o;
    }
  }
  __syncthreads();
}
extern __shared__ float lds[];
__global__ void __launch_bounds__(64, 3) bench(const float* A, float* out, float* Wout, long long* cyc, int reps) {
  constexpr int N = 30, hs = 36;
  float* H = lds; float* x = lds + 32 * hs;
  long long t0 = 0, total = 0;
  for (int r = 0; r < reps; r++) {
    for (int w = LANE; w < 31 * hs; w += 64) H[w] = A[w];
    __syncthreads();
    t0 = __builtin_readcyclecounter();
    chol_mfma<N>(H, hs, x);
    total += __builtin_readcyclecounter() - t0;
  }
  if (LANE < N) out[blockIdx.x * 32 + LANE] = x[LANE];
  if (blockIdx.x == 0) for (int w = LANE; w < 30 * hs; w += 64) Wout[w] = H[w];
  if (LANE == 0) cyc[blockIdx.x] = total / reps;
}
int main() {
  const int N = 30, hs = 36;
  std::vector<float> A(31 * hs, 0.f);
  std::vector<double> Hd(N * N), g(N);
  srand(1);
  // SPD: B B' + diag, lower triangle stored
  std::vector<double> B(N * N);
  for (auto& b : B) b = (rand() / (double)RAND_MAX - 0.5);
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < N; k++) s += B[i * N + k] * B[j * N + k]; Hd[i * N + j] = s + (i == j ? 1.0 + 0.1 * i : 0.0); }
  for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) A[i * hs + j] = (float)Hd[i * N + j];
  for (int j = 0; j < N; j++) { g[j] = 1.0 + 0.1 * j * ((j & 1) ? -1 : 1); A[N * hs + j] = (float)g[j]; }
  // reference: gaussian elimination in double
  std::vector<double> M(Hd), xr(g);
  for (int j = 0; j < N; j++) { for (int i = j + 1; i < N; i++) { double f = M[i * N + j] / M[j * N + j]; for (int k = j; k < N; k++) M[i * N + k] -= f * M[j * N + k]; xr[i] -= f * xr[j]; } }
  for (int i = N - 1; i >= 0; i--) { double s = xr[i]; for (int k = i + 1; k < N; k++) s -= M[i * N + k] * xr[k]; xr[i] = s / M[i * N + i]; }
  float *dA, *dout, *dW; long long* dc;
  hipMalloc(&dA, A.size() * 4); hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  const int maxb = 256 * 12;
  hipMalloc(&dout, maxb * 32 * 4); hipMalloc(&dc, maxb * 8); hipMalloc(&dW, 30 * hs * 4);
  for (int perCU : {1, 4, 8, 12}) {
    int nb = 256 * perCU; size_t ldsb = 12656;
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(bench, dim3(nb), dim3(64), ldsb, 0, dA, dout, dW, dc, 200); hipDeviceSynchronize(); }
    std::vector<long long> c(nb); hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= nb;
    std::vector<float> o(32); hipMemcpy(o.data(), dout, 32 * 4, hipMemcpyDeviceToHost);
    double err = 0, nrm = 0; for (int i = 0; i < N; i++) { err = fmax(err, fabs(o[i] - xr[i])); nrm = fmax(nrm, fabs(xr[i])); }
    printf("mfma waves/CU %2d: %.0f cycles per call; max |x - x_ref| = %.3g (|x| max %.3g)  x[0..2] %g %g %g ref %g %g %g\n", perCU, m, err, nrm, o[0], o[1], o[2], xr[0], xr[1], xr[2]);
  }
  // W check: x2 = W (W' g)
  std::vector<float> W(30 * hs); hipMemcpy(W.data(), dW, W.size() * 4, hipMemcpyDeviceToHost);
  std::vector<double> y(N, 0.0), x2(N, 0.0);
  for (int k = 0; k < N; k++) for (int m = 0; m < N; m++) y[k] += W[m * hs + k] * g[m];
  for (int m = 0; m < N; m++) for (int k = 0; k < N; k++) x2[m] += W[m * hs + k] * y[k];
  double err = 0; for (int i = 0; i < N; i++) err = fmax(err, fabs(x2[i] - xr[i]));
  printf("reuse check: max |W W' g - x_ref| = %.3g\n", err);
  return 0;
}
