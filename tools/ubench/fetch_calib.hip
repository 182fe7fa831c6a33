// FETCH_SIZE / WRITE_SIZE calibration (round 6): launches with a KNOWN number of bytes moved, by access pattern, so that the PMC figures of
// rb_step_kernel (whose traffic is dword loads / stores of scratch-row arrays, not wide streaming) can be read with a factor measured on ITS pattern.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f --output-format csv -- ./fetch_calib      (and once more with WRITE_SIZE)
// Every kernel touches each byte of its region exactly once in a 2 GiB buffer (nothing is re-read, nothing fits a cache); the program prints the bytes each launch
// requests and the bytes of the 64-byte lines it touches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// wide: one float4 per lane, consecutive (what the guide's x2 correction was calibrated on)
__global__ void read_wide(const float4* p, float* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0;
  for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
// dword, coalesced: a wave reads 64 consecutive dwords (256 B) -- a PFOR over a scratch-row array
__global__ void read_dword(const float* p, float* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
// dword, row-strided: wave w reads 64 consecutive dwords of "array a" in "env row e": segments of 256 B that are 282 kB apart between consecutive waves (rb_step_kernel's
// scratch rows: one 282-kB row per env, a wave per env)
__global__ void read_rowstrided(const float* p, float* out, int nrows, int nseg, size_t row_words) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63; float acc = 0;
  if (wave < nrows) for (int s = 0; s < nseg; s++) acc += p[(size_t)wave * row_words + (size_t)s * 1024 + lane];
  if (acc == 12345.678f) out[0] = acc;
}
// dword, scattered: every lane reads ONE dword of its own 128-byte-aligned line (lists, index chains)
__global__ void read_scattered(const float* p, float* out, size_t nlines) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float acc = 0;
  for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 32];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void write_wide(float4* p, size_t n4) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4); }
__global__ void write_dword(float* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f; }
__global__ void write_rowstrided(float* p, int nrows, int nseg, size_t row_words) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave < nrows) for (int s = 0; s < nseg; s++) p[(size_t)wave * row_words + (size_t)s * 1024 + lane] = 1.f;
}
int main() {
  const size_t bytes = (size_t)2 << 30;
  float *buf, *out; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const size_t region = (size_t)1 << 30;   // 1 GiB per launch
  const int grid = 256 * 16, blk = 256;
  const int nrows = 4096, nseg = 64; const size_t row_words = 70656;   // 282 kB rows; 64 segments of 256 B per row: 4096 x 64 x 256 B = 64 MiB requested
  printf("launch                 requested_bytes   bytes_of_64B_lines_touched\n");
  hipLaunchKernelGGL(read_wide, dim3(grid), dim3(blk), 0, 0, (const float4*)buf, out, region / 16); CHECK(hipDeviceSynchronize());
  printf("read_wide              %zu   %zu\n", region, region);
  hipLaunchKernelGGL(read_dword, dim3(grid), dim3(blk), 0, 0, buf + region / 4, out, region / 4); CHECK(hipDeviceSynchronize());
  printf("read_dword             %zu   %zu\n", region, region);
  hipLaunchKernelGGL(read_rowstrided, dim3(nrows * 64 / blk), dim3(blk), 0, 0, buf, out, nrows, nseg, row_words); CHECK(hipDeviceSynchronize());
  printf("read_rowstrided        %zu   %zu\n", (size_t)nrows * nseg * 256, (size_t)nrows * nseg * 256);
  hipLaunchKernelGGL(read_scattered, dim3(grid), dim3(blk), 0, 0, buf + region / 4, out, region / 128 / 4); CHECK(hipDeviceSynchronize());
  printf("read_scattered         %zu   %zu\n", region / 128 / 4 * 4, region / 128 / 4 * 64);
  hipLaunchKernelGGL(write_wide, dim3(grid), dim3(blk), 0, 0, (float4*)buf, region / 16); CHECK(hipDeviceSynchronize());
  printf("write_wide             %zu   %zu\n", region, region);
  hipLaunchKernelGGL(write_dword, dim3(grid), dim3(blk), 0, 0, buf + region / 4, region / 4); CHECK(hipDeviceSynchronize());
  printf("write_dword            %zu   %zu\n", region, region);
  hipLaunchKernelGGL(write_rowstrided, dim3(nrows * 64 / blk), dim3(blk), 0, 0, buf, nrows, nseg, row_words); CHECK(hipDeviceSynchronize());
  printf("write_rowstrided       %zu   %zu\n", (size_t)nrows * nseg * 256, (size_t)nrows * nseg * 256);
  return 0;
}
