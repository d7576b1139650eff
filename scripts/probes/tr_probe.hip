// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds u16 values = their own element index; every lane reads
// 8 bytes at a caller-given byte address; out[lane][0..3] = the four 16-bit values it received.
#include <hip/hip_runtime.h>
extern "C" __global__ void tr_probe(const int* addr, unsigned short* out, int n16) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < n16; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
extern "C" int run_tr(const void* addr, void* out, int n16, void* stream) {
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)addr, (unsigned short*)out, n16);
  return (int)hipGetLastError();
}
