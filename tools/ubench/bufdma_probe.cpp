// Probe of `buffer_load_dwordx4 ... offen lds` on gfx950 (the LDS-DMA form with an SGPR resource descriptor + SGPR offset + per-lane 32-bit
// offset): where do the bytes land (M0 above 64 KiB), what do out-of-range lanes write, is the SGPR offset part of the range check?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/bufdma_probe.cpp -o tools/ubench/bufdma_probe.bin && tools/ubench/bufdma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void probe(const unsigned* src, unsigned nbytes, unsigned soff, unsigned* out, unsigned lds_base, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 40960; i += 64) smem[i] = 0xFFFFFFFFu;  // 160 KiB of ones
  __syncthreads();
  u32x4 rsrc;
  const uint64_t p = (uint64_t)src;
  rsrc[0] = (unsigned)p;
  rsrc[1] = (unsigned)(p >> 32) & 0xFFFFu;  // stride 0
  rsrc[2] = nbytes;                         // num_records (bytes for a raw buffer)
  rsrc[3] = 0x00020000u;
  // lane l asks for the vector at byte offset: mode 0: 16 l;  mode 1: lanes >= 32 out of range by voffset;  mode 2: 16 l with soff pushing the
  // upper lanes out of range
  unsigned voff = 16u * lane;
  if (mode == 1 && lane >= 32) voff = 0x80000000u + 16u * lane;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "memory");
  __syncthreads();
  for (int i = lane; i < 256 + 8; i += 64) out[i] = smem[lds_base / 4 - 4 + i];  // 16 bytes before the destination .. 16 bytes behind the 1 KiB
}

int main() {
  const int N = 4096;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x1000u + i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4);
  hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct { const char* name; unsigned nbytes, soff, lds, mode; } cases[] = {
      {"plain, LDS base 4 KiB", N * 4u, 0, 4096, 0},
      {"plain, LDS base 100 KiB (M0 > 64 KiB)", N * 4u, 0, 100 * 1024, 0},
      {"soffset 256 B", N * 4u, 256, 100 * 1024, 0},
      {"lanes 32.. out of range by voffset", N * 4u, 0, 100 * 1024, 1},
      {"num_records 1 KiB + soffset 512: lanes 32.. beyond the records", 1024u, 512, 100 * 1024, 2},
  };
  for (auto& c : cases) {
    hipMemset(o, 0, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, c.nbytes, c.soff, o, c.lds, (int)c.mode);
    std::vector<unsigned> r(264);
    if (hipMemcpy(r.data(), o, 264 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: copy failed\n", c.name); continue; }
    printf("%s\n  before: %08x  lane0: %08x %08x %08x %08x  lane1: %08x ..  lane31: %08x  lane32: %08x %08x  lane63: %08x .. %08x  behind: %08x\n", c.name, r[3], r[4], r[5], r[6],
           r[7], r[8], r[4 + 31 * 4], r[4 + 32 * 4], r[4 + 32 * 4 + 1], r[4 + 63 * 4], r[4 + 63 * 4 + 3], r[4 + 256]);
  }
  printf("expect plain: lane l holds words 0x1000 + 4 l .. + 3; soffset 256: + 64 words; out-of-range lanes: 00000000 (or ffffffff = nothing written)\n");
  return 0;
}
