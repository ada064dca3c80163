// Calibration of rocprofv3's FETCH_SIZE on the access pattern of k_conv3x3p's halo requests: LDS-DMA
// (buffer_load ... lds, 16 bytes per lane) of ROWS bytes per pixel out of a pixel-major tensor with a 608-byte pixel stride
// (304 bf16 channels), against the same number of bytes read contiguously.  Build + run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./probe_fetch        (tools/probe_fetch.sh)
// Kernels (each reads every pixel of an 8 x 256 x 416 image once):
//   k_rows<64>(chunk c)   : bytes [64 c, 64 c + 64) of every pixel  -- one channel chunk of the conv (useful 54.5 MB)
//   k_rows<128>(chunk c)  : bytes [128 c, 128 c + 128)              -- what a 64-channel chunk would ask for (109 MB)
//   k_contig              : 54.5 MB / 109 MB contiguous
//   k_rows<64> chunk 0 then chunk 1 back to back: does the second half of the lines come from L2?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int ROWB>
__global__ __launch_bounds__(256) void k_rows(const char* x, long long npix, int stride, int chunk, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int LPR = ROWB / 16, PPI = 64 / LPR;           // lanes per row, pixels per instruction
  const long long per_wg = (npix + gridDim.x - 1) / gridDim.x;
  long long p0 = (long long)blockIdx.x * per_wg, p1 = p0 + per_wg;
  if (p1 > npix) p1 = npix;
  for (long long base = p0 + wv * PPI; base < p1; base += 4 * PPI) {
    const long long p = base + l / LPR;
    const unsigned long long addr = (unsigned long long)(x + p * stride + chunk * ROWB);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + base * stride), 0, 0x7fffffff, 0x00020000);
    const unsigned off = p < p1 ? (unsigned)((p - base) * stride + chunk * ROWB + (l % LPR) * 16) : 0x80000000u;
    (void)addr;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + wv * 1024), 16, off, 0, 0, 0);
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[5] == 77) sink[0] = 1;
}

__global__ __launch_bounds__(256) void k_contig(const char* x, long long bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long per_wg = ((bytes / 1024 + gridDim.x - 1) / gridDim.x) * 1024;
  long long b0 = (long long)blockIdx.x * per_wg, b1 = b0 + per_wg;
  if (b1 > bytes) b1 = bytes;
  for (long long base = b0 + wv * 1024; base < b1; base += 4096) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + base), 0, 1024, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + wv * 1024), 16, (unsigned)(l * 16), 0, 0, 0);
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[5] == 77) sink[0] = 1;
}

int main() {
  const long long npix = 8ll * 256 * 416;
  const int stride = 608;
  char* x; unsigned* sink;
  hipMalloc(&x, npix * stride + 4096); hipMalloc(&sink, 64);
  hipMemset(x, 1, npix * stride + 4096);
  char* flush; hipMalloc(&flush, 600ll << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](const char* name, auto launch) {
    hipMemset(flush, 2, 600ll << 20);                      // evict L2 / Infinity Cache between experiments
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us\n", name, ms * 1e3);
  };
  const int G = 1024;
  timed("rows64 chunk0 (54.5 MB useful)", [&] { k_rows<64><<<G, 256>>>(x, npix, stride, 0, sink); });
  timed("rows64 chunk1", [&] { k_rows<64><<<G, 256>>>(x, npix, stride, 1, sink); });
  timed("rows64 chunk0 + chunk1 back to back", [&] { k_rows<64><<<G, 256>>>(x, npix, stride, 0, sink); k_rows<64><<<G, 256>>>(x, npix, stride, 1, sink); });
  timed("rows128 chunk0 (109 MB useful)", [&] { k_rows<128><<<G, 256>>>(x, npix, stride, 0, sink); });
  timed("contiguous 54.5 MB", [&] { k_contig<<<G, 256>>>(x, npix * 64, sink); });
  timed("contiguous 109 MB", [&] { k_contig<<<G, 256>>>(x, npix * 128, sink); });
  timed("contiguous 518 MB (whole tensor)", [&] { k_contig<<<G, 256>>>(x, npix * stride, sink); });
  return 0;
}
