// SAP neighbour sum on the cell list (reference src/sap.rs:155-204).  Included by kernels.hip inside namespace arp.
//
// For every side-chain atom x: the f32 sum, over the side-chain atoms y within sap_radius of x (x itself included, inclusive test in
// f64 as rstar's locate_within_distance), of weight(y) = hydrophobicity(resn(y)) * clamp(sasa(y) / max_sc_asa(resn(y)), 0, 1).  Same
// access pattern as the contact search, different reduction: the grid is built over the side-chain atoms only (everything else
// carries the "not in the grid" attribute bit) and ORDERED (slots inside a cell follow the atom index), so that every atom adds up its
// neighbours in an order that is a function of the input only and the f32 sum is reproducible bit for bit.
//
// Round 4: the emit kernel's data flow instead of one thread per atom gathering 48-byte records.  One block per 64 home slots, lane = home
// atom; the nine x-contiguous slot windows of the full shell (cells are x-major, 2 kx + 1 of them per window); per window the covering
// slot interval is staged through LDS in 128-record chunks as {x, y, z relative to the box midpoint in f32, weight} -- the weight gathered
// by the original index once per staged record instead of once per test -- and every lane walks its own part of the chunk with one
// ds_read_b128 and ~10 vector instructions per test.  The decision is made in f32 wherever the f32 distance is farther from r^2 than the
// prefilter margin of the grid build (DESIGN.md "Prefilter margin": a proven bound on what the f32 records can be off by); the few tests
// inside that band gather the f64 coordinates and decide exactly, behind a wave-uniform branch.  No compaction: nothing is emitted.
// A task's nine windows go to the nine waves of ONE block (a task is a chain of dependent round trips -- window bounds, staging, the
// weight gather behind the index load -- and a 10^6-atom input has only seven tasks per SIMD: one wave per task ran at the latency of the
// chain, 62 us; measured profiles/r04_sap.txt).  Each wave adds up its window in slot order, the block adds the nine partial sums in window
// order: still a function of the input only.
// SPLIT = waves per task (9 or 3: one window, or one z layer of three windows per wave; 1 = the whole shell, measured and never the best):
// the split shortens the chain but every wave of a task loads the home records and reduces its window bounds again, so large inputs --
// enough tasks to fill the chip -- take fewer waves per task (launch_neighbor_sum).
constexpr uint32_t kSapChunk = 128, kSapWindows = 9, kSapGroup = 4;
struct SapWaveLds { float4 rec[kSapChunk + kSapGroup]; };

template <uint32_t SPLIT>
__global__ __launch_bounds__(SPLIT * 64) void k_neighbor_sum(const GridParams *gp, const uint32_t *cell_start, Sorted so, double r2, const float *weight, float *out) {
    __shared__ SapWaveLds wl[SPLIT];
    __shared__ float partial[SPLIT][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    SapWaveLds &w = wl[wave];
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, kx = gp->kx, n_heavy = gp->n_heavy;
    const uint32_t sy = gp->sy_shift;
    const uint32_t slot0 = blockIdx.x * 64u;
    if (slot0 >= n_heavy) return;  // (whole blocks only)
    const uint32_t a = slot0 + lane;
    const bool have = a < n_heavy;
    // f32 decision bounds: sure inside below r2 - margin, sure outside above r2 + margin (the margin covers the storage rounding of the
    // records; the arithmetic below adds a few ulps of d^2, orders of magnitude less)
    const float margin = gp->prefilter_margin;
    const float lo_thr = __double2float_rd(r2 - (double)margin - 1e-6 * r2), hi_thr = __double2float_ru(r2 + (double)margin + 1e-6 * r2);
    float hx = 0.f, hy = 0.f, hz = 0.f;
    double ax = 0.0, ay = 0.0, az = 0.0;
    constexpr uint32_t kPer = kSapWindows / SPLIT;  // windows of this wave: k = wave * kPer .. + kPer
    uint32_t wlo[kPer], whi[kPer], orig = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) wlo[q] = whi[q] = 0u;
    if (have) {
        const float4 h = so.rec[a];
        hx = h.x; hy = h.y; hz = h.z;
        const Fat &f = fat_at<false>(so.fat, a);
        ax = f.x; ay = f.y; az = f.z; orig = f.orig;
        const uint32_t c = f.cell, row = c / nx, cx = c - row * nx;
        uint32_t cy, cz;
        grid_row_decode(row, ny, nzt, sy, cy, cz);
        const uint32_t xlo = cx > kx ? cx - kx : 0u, xhi = min(cx + kx, nx - 1u);
#pragma unroll
        for (uint32_t q = 0; q < kPer; q++) {  // all window bounds up front (independent loads)
            const uint32_t k = wave * kPer + q;
            const int zz = (int)cz + (int)(k / 3u) - 1, yy = (int)cy + (int)(k % 3u) - 1;
            if (zz >= 0 && zz < (int)nzt && yy >= 0 && yy < (int)ny) {
                const uint32_t r = grid_row((uint32_t)yy, (uint32_t)zz, ny, nzt, sy) * nx;
                wlo[q] = cell_start[r + xlo]; whi[q] = cell_start[r + xhi + 1u];
            }
        }
    }
    float acc = 0.0f;
#pragma unroll 1
    for (uint32_t q = 0; q < kPer; q++) {
        uint32_t lo = wlo[0], hi = whi[0];
#pragma unroll
        for (uint32_t j = 1; j < kPer; j++) if (q == j) { lo = wlo[j]; hi = whi[j]; }
        const bool nonempty = lo < hi;
        const uint32_t Lw = wave_min_u32(nonempty ? lo : 0xFFFFFFFFu), Hw = wave_max_u32(nonempty ? hi : 0u);
#pragma unroll 1
        for (uint32_t cs = Lw; cs < Hw; cs += kSapChunk) {
            const uint32_t ce = min(cs + kSapChunk, Hw);
            const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
            const uint32_t len = (nonempty && j1 > j0) ? j1 - j0 : 0u;
            if (!__any(len != 0u)) {  // no lane's window reaches into this chunk: on to the first slot any lane still needs (lanes of one task can sit in
                // rows whose neighbour rows lie far apart in the cell order -- a whole y strip apart at a strip's edge, arp_internal.h grid_row)
                const uint32_t need = wave_min_u32((nonempty && hi > ce) ? max(lo, ce) : 0xFFFFFFFFu);
                if (need >= Hw) break;
                cs = need - kSapChunk;  // (>= cs: the chunk was a whole one, or no lane would be left; the loop's increment follows)
                continue;
            }
            wave_lds_fence();  // previous chunk fully consumed
            {   // stage: {x, y, z, weight of the record's atom}; clamped addresses (the pad of a short chunk is never inside a window)
                const uint32_t p0 = min(cs + lane, ce - 1u), p1 = min(cs + lane + 64u, ce - 1u);
                const float4 r0 = so.rec[p0], r1 = so.rec[p1];
                const uint32_t o0 = fat_at<false>(so.fat, p0).orig, o1 = fat_at<false>(so.fat, p1).orig;
                const float w0 = weight[o0], w1 = weight[o1];
                w.rec[lane] = make_float4(r0.x, r0.y, r0.z, w0); w.rec[lane + 64u] = make_float4(r1.x, r1.y, r1.z, w1);
            }
            wave_lds_fence();
            const uint32_t off = len ? j0 - cs : 0u;
            const float4 *win = w.rec + off;
#pragma unroll 1
            for (uint32_t it0 = 0; __any(it0 < len); it0 += kSapGroup) {
                float d2[kSapGroup], wt[kSapGroup];
#pragma unroll
                for (uint32_t u = 0; u < kSapGroup; u++) {  // (a lane past its window end reads on -- other records, another wave's buffer, at worst past the block's
                                                            // LDS allocation, which never faults (tests/lds_oob) -- and the value is dropped below: `mine`)
                    const float4 r = win[it0 + u];
                    const float dx = r.x - hx, dy = r.y - hy, dz = r.z - hz;
                    d2[u] = __fmaf_rn(dx, dx, __fmaf_rn(dy, dy, dz * dz));
                    wt[u] = r.w;
                }
#pragma unroll
                for (uint32_t u = 0; u < kSapGroup; u++) {
                    const bool mine = it0 + u < len;
                    bool in = mine & (d2[u] <= lo_thr);
                    const bool band = mine & !in & (d2[u] <= hi_thr);
                    if (__any(band)) {  // rare: the f32 distance cannot decide -- the reference's own test, f64, inclusive (sap.rs:183, rstar)
                        if (band) {
                            const Fat &b = fat_at<false>(so.fat, cs + off + it0 + u);
                            in = sq_dist(ax, ay, az, b.x, b.y, b.z) <= r2;
                        }
                    }
                    acc += in ? wt[u] : 0.0f;  // (adding 0.0f is exact: the sum runs over the neighbours in window, then slot order)
                }
            }
        }
    }
    partial[wave][lane] = acc;
    __syncthreads();
    if (wave == 0u && have) {
        float sum = 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < SPLIT; k++) sum += partial[k][lane];  // window order
        out[orig] = sum;
    }
}

// tasks below which a task's windows go to nine waves instead of three (profiles/r04_sap.txt: sum kernel on S1 clouds, 9 / 3 / 1 waves per
// task: 3 x 10^4 atoms 9.4 / 15.7 / 25.1 us, 10^5 13.4 / 16.1 / 26.8, 3 x 10^5 30.3 / 23.0 / 32.9, 10^6 80.8 / 56.3 / 80.8)
constexpr uint32_t kSapSplit9Below = 3072;
void launch_neighbor_sum(const DevAtoms &in, const Workspace &ws, double radius, double r2, const float *weight, float *out, hipStream_t st, Profiler *prof) {
    launch_grid(in, ws, st, prof, radius, /* ordered: slots follow the atom index inside a cell */ true);
    if (prof) prof->begin("sap_sum", st);
    const uint32_t tasks = (in.n + 63u) / 64u;  // (an upper bound: only the side-chain atoms are in the grid; blocks beyond them return at once)
#define ARP_LAUNCH_SAP(S) hipLaunchKernelGGL(k_neighbor_sum<S>, dim3(tasks), dim3(S * 64), 0, st, (const GridParams *)ws.grid, (const uint32_t *)ws.cell_start, ws.sorted, r2, weight, out)
    if (in.n) {
        if (tasks < kSapSplit9Below) ARP_LAUNCH_SAP(9);
        else ARP_LAUNCH_SAP(3);
    }
#undef ARP_LAUNCH_SAP
    if (prof) prof->end(st);
}
