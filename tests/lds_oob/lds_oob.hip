// Test-only (not part of the product): the hardware contract k_emit's prefilter and the SAP sum lean on.  A lane whose slot window is exhausted
// keeps reading 16-byte records at immediate offsets from its window start and its result bits are dropped afterwards (pairs_emit.inl, "Phase 1,
// one run"; sap.inl) -- the reads may run past the wave's arrays and, for the last wave of a block, past the block's LDS allocation.  What the
// kernels need is that such a read NEVER FAULTS; the value is discarded.  Measured here (round 5; the first version of this test expected zeros
// everywhere and failed): an LDS read past the workgroup's allocation raises nothing; just past the end it returns whatever the padding of the
// allocation granule holds (stale words of earlier kernels -- so nothing may ever depend on the value), far past it returns 0.  This kernel reads
// past a 4 KB allocation on purpose (inline assembly: a C++ out-of-bounds access would be undefined behaviour for the compiler to exploit); the
// host checks that the in-range control reads saw the fill pattern, that reads 64 KB and more past the end came back 0, and that the launch
// completed without a fault; it reports where the nonzero words stop.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kWords = 1024;  // 4 KB: the whole LDS allocation of a workgroup of this kernel

__global__ __launch_bounds__(256) void k_lds_oob(u32x4 *out, const uint32_t *offsets, uint32_t n_off) {
    __shared__ uint32_t lds[kWords];
    for (uint32_t k = threadIdx.x; k < kWords; k += blockDim.x) lds[k] = 0xA5A5A5A5u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
    for (uint32_t j = 0; j < n_off; j++) {
        const uint32_t addr = base + offsets[j] + 16u * threadIdx.x;
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        out[(blockIdx.x * n_off + j) * blockDim.x + threadIdx.x] = v;
    }
}

// Returns 0 when every expectation held, else the number of words that did not; -1 on a HIP error.  offsets[0] = 0 is the in-range control.
extern "C" int lds_oob_check(int verbose) {
    const std::vector<uint32_t> offsets = {0u, 4096u, 4096u + 4080u, 16384u, 65536u - 16u, 65536u, 81920u, 131072u, 163840u - 4096u, 163840u, 1u << 20, 0x7FFFF000u};
    const uint32_t n_off = (uint32_t)offsets.size(), blocks = 64, threads = 256;
    uint32_t *d_off = nullptr;
    u32x4 *d_out = nullptr;
    const size_t n_out = (size_t)blocks * n_off * threads;
    if (hipMalloc((void **)&d_off, n_off * 4) != hipSuccess || hipMalloc((void **)&d_out, n_out * 16) != hipSuccess) return -1;
    if (hipMemcpy(d_off, offsets.data(), n_off * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemset(d_out, 0xFF, n_out * 16) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_lds_oob, dim3(blocks), dim3(threads), 0, nullptr, d_out, (const uint32_t *)d_off, n_off);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint32_t> h(n_out * 4);
    if (hipMemcpy(h.data(), d_out, n_out * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipFree(d_off); (void)hipFree(d_out);
    int bad = 0;
    uint64_t last_nonzero = 0;  // largest out-of-range byte offset that read back nonzero (the extent of the allocation granule's padding)
    for (uint32_t b = 0; b < blocks; b++)
        for (uint32_t j = 0; j < n_off; j++)
            for (uint32_t t = 0; t < threads; t++)
                for (uint32_t w = 0; w < 4; w++) {
                    const uint32_t got = h[(((size_t)b * n_off + j) * threads + t) * 4 + w];
                    const uint64_t byte = (uint64_t)offsets[j] + 16u * t + 4u * w;  // offset of this word from the start of the allocation
                    const bool in_range = byte < 4u * kWords, far = byte >= 4u * kWords + 65536u;
                    if (!in_range && got != 0u && byte > last_nonzero) last_nonzero = byte;
                    const bool ok = in_range ? got == 0xA5A5A5A5u : (far ? got == 0u : true);  // (just past the end: unspecified, only "no fault")
                    if (!ok) { if (verbose && bad < 8) fprintf(stderr, "block %u offset %u lane %u word %u: %08x\n", b, offsets[j], t, w, got); bad++; }
                }
    if (verbose) fprintf(stderr, "lds_oob: allocation 4096 B; nonzero words read up to byte offset %llu past the start\n", (unsigned long long)last_nonzero);
    return bad;
}
