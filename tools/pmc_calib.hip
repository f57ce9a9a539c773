// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this part for the access widths the step's kernels use (MI355X_MICROARCH.md,
// HBM: "FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel moves exactly N bytes (256 MiB: beyond the 256 MiB
// infinity cache together with the other array); tools/pmc_calib.sh collects the two counters per kernel.
//   read16 / read4       coalesced reads, 16 / 4 bytes per lane
//   read4_rows64         4 bytes per lane, a wavefront touches four 64-byte runs 4 KB apart (the 16 x 16 tiles' operand loads)
//   write16 / write4     coalesced writes
//   write_rows64         64-byte row segments 1 KB apart (a 16 x 16 tile's Adam stores), write_rows128: 128-byte segments
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void read16(const f32x4* p, size_t n, float* sink) { f32x4 a = {0, 0, 0, 0}; for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) a += p[i]; if (a[0] + a[1] + a[2] + a[3] == 1.2345f) *sink = 1; }
__global__ void read4(const float* p, size_t n, float* sink) { float a = 0; for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) a += p[i]; if (a == 1.2345f) *sink = 1; }
__global__ void read4_rows64(const float* p, size_t n, float* sink) {      // lane (li, lc): element (row 4 g + lc) * 1024 + 16 h + li: 64-byte runs of rows 4 KB apart
  float a = 0; const int lane = threadIdx.x & 63, li = lane & 15, lc = lane >> 4; const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, nw = gridDim.x * 4ull;
  const size_t rows = n / 1024;
  for (size_t u = wave; u < (rows / 4) * 64; u += nw) { const size_t g = u / 64, h = u % 64; a += p[(4 * g + lc) * 1024 + 16 * h + li]; }
  if (a == 1.2345f) *sink = 1;
}
__global__ void write16(f32x4* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = f32x4{1, 2, 3, 4}; }
__global__ void write4(float* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = 1.f; }
template <int SEG>      // floats per row segment (16: 64 bytes, 32: 128 bytes); rows 256 floats (1 KB) apart inside a tile of 16 rows
__global__ void write_rows(float* p, size_t n) {
  const size_t tiles = n / (16 * 256) * (256 / SEG);      // tiles of 16 rows x SEG columns covering the array
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const size_t band = t / (256 / SEG), col = (t % (256 / SEG)) * SEG;
    for (int e = threadIdx.x; e < 16 * SEG; e += 256) { const int r = e / SEG, c = e % SEG; p[(band * 16 + r) * 256 + col + c] = 2.f; }
  }
}
int main() {
  const size_t bytes = 256ull << 20, n = bytes / 4;
  float *a, *b, *sink; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    write4<<<4096, 256>>>(b, n);          // (evicts a from the caches between the read kernels)
    read16<<<4096, 256>>>((const f32x4*)a, n / 4, sink);
    write16<<<4096, 256>>>((f32x4*)b, n / 4);
    read4<<<4096, 256>>>(a, n, sink);
    write4<<<4096, 256>>>(b, n);
    read4_rows64<<<4096, 256>>>(a, n, sink);
    write_rows<16><<<4096, 256>>>(b, n);
    read16<<<4096, 256>>>((const f32x4*)a, n / 4, sink);
    write_rows<32><<<4096, 256>>>(b, n);
  }
  hipDeviceSynchronize();
  printf("moved %zu bytes per kernel\n", bytes);
  return 0;
}
