#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned int* p, unsigned int* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeef;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  // lane l loads 16 bytes from a permuted source granule; odd lanes are out of bounds
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : (unsigned)((63 - threadIdx.x) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, (unsigned)(threadIdx.x * 16), 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(256), o(512);
  for (int i = 0; i < 256; ++i) h[i] = i;
  unsigned *d, *dout; hipMalloc(&d, 1024); hipMalloc(&dout, 2048);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, dout, 1024); hipDeviceSynchronize();
  hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    unsigned exp = (l & 1) ? 0u : (unsigned)((63 - l) * 4 + j);
    if (o[l * 4 + j] != exp) { if (bad < 8) printf("lane %d j %d got %08x exp %08x\n", l, j, o[l*4+j], exp); ++bad; }
    if (o[256 + l * 4 + j] != (unsigned)(l * 4 + j)) ++bad;
  }
  printf("buffer_load_lds: lane-linear dest + per-lane source + OOB->0: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  return 0;
}
