// Calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access patterns of psa_enum_wave:
// wave-uniform scalar loads of 64 bytes, and one dword per lane (256 bytes per wave).
// Each kernel streams a buffer once; compare FETCH_SIZE with the bytes printed.
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib [MiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

// every wave reads its own contiguous piece with s_load_dwordx16
__global__ __launch_bounds__(64) void scalar_stream(const int *__restrict__ p, size_t ints_per_wave, int *out) {
  const int *q = p + (size_t)blockIdx.x * ints_per_wave;
  int acc = 0;
  for (size_t i = 0; i < ints_per_wave; i += 16) {
    int g[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = q[i + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc ^= g[k];
  }
  if (acc == 0x7fffffff) out[0] = acc;
}

// every wave reads its own contiguous piece, one dword per lane and step
__global__ __launch_bounds__(64) void dword_stream(const int *__restrict__ p, size_t ints_per_wave, int *out) {
  const int *q = p + (size_t)blockIdx.x * ints_per_wave;
  int acc = 0;
  for (size_t i = threadIdx.x; i < ints_per_wave; i += 64) acc ^= q[i];
  if (acc == 0x7fffffff) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? (size_t)std::atoi(argv[1]) : 1024;
  const size_t bytes = mib << 20, ints = bytes / 4;
  const int waves = 16384;
  const size_t per = ints / waves / 16 * 16;
  int *d, *out;
  CK(hipMalloc((void **)&d, bytes));
  CK(hipMalloc((void **)&out, 4));
  CK(hipMemset(d, 1, bytes));
  hipLaunchKernelGGL(scalar_stream, dim3(waves), dim3(64), 0, 0, d, per, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(dword_stream, dim3(waves), dim3(64), 0, 0, d, per, out);
  CK(hipDeviceSynchronize());
  std::printf("each kernel read %zu bytes (%.1f KB)\n", per * 4 * waves, per * 4.0 * waves / 1024);
  return 0;
}
