// The default single-pass emitter: the HOME operand of the exact phase lives in LDS.  Included by kernels.hip after pairs_lds.inl,
// inside namespace arp.
//
// k_pairs<kEmit> (pairs.inl) gathers two 48-byte records per survivor, each a 64-byte line out of the L2: 3.4 GB per launch on the
// headline input = 16 TB/s, the rate MI355X_MICROARCH.md measures for L2-served gathers.  Half of those records are the task's own 64
// home atoms, fetched again for every batch.  Here the wave keeps them in LDS for the duration of the task (40 bytes each) and phase 2
// reads the home side with three ds_read; queue entries shrink to 4 bytes (home lane << 26 | neighbour slot); because the home records
// change with the task the queue is drained at every task end.
//
// What it took to make this pay (profiles/r02_emit_kernels.txt): all these kernels are paced by the chain of dependent round trips of a
// wave times the resident waves, so the 2.5 KB of home records per wave had to come out of the staged chunk (160 records instead of 256)
// and the decision tables had to be shared by more waves (12-wave blocks, two per CU) to keep 6 waves per SIMD: 199 us against 210 us
// for k_pairs (S1 181 / 188, the config-5 pack 864 / 922, contacts-only 85 / 92).  At 4 waves per SIMD the same kernel ran 250 us, 228 us
// with kHMulti = 3 batches behind one gather round trip (the second and third batch in flight cost ten registers each, which 6 waves
// do not have).
#ifndef ARP_H_WAVES
#define ARP_H_WAVES 12
#endif
#ifndef ARP_H_BLOCKS_PER_CU
#define ARP_H_BLOCKS_PER_CU 2
#endif
#ifndef ARP_H_CHUNK
#define ARP_H_CHUNK 160
#endif
constexpr uint32_t kHChunk = ARP_H_CHUNK;     // staged neighbour records per chunk
constexpr int kHWaves = ARP_H_WAVES;         // waves per block (7-8 KB of LDS each + the block's 8.4 KB of decision tables)
constexpr uint32_t kHBlocks = 256u * ARP_H_BLOCKS_PER_CU;
constexpr int kHWavesPerSimd = (ARP_H_WAVES * ARP_H_BLOCKS_PER_CU) / 4;
constexpr uint32_t kHSlotBits = 26;          // neighbour slot bits of a queue entry; the launcher routes larger inputs elsewhere
constexpr uint32_t kHMaxSlots = (1u << kHSlotBits) - 64u;
#ifndef ARP_H_MULTI
#define ARP_H_MULTI 1
#endif
constexpr uint32_t kHMulti = ARP_H_MULTI;     // batches classified per round trip (their gathers are in flight together)
constexpr uint32_t kHQueue = 64u * kHMulti + 64u;  // the queue fills to 64 kHMulti (+ up to 63 of the last round)

struct WaveLdsH {
    float4 nrec[kHChunk + kBlock];           // f32 prefilter records of the staged chunk (+ kBlock: over-reads stay in bounds)
    u32x4 hxy[64]; u32x4 hzm[64]; u32x2 hco[64];   // the task's home atoms: {x, y}, {z, pw, res_ord}, {crm, orig}
    uint32_t queue[kHQueue];                 // phase-1 survivors: home lane << 26 | neighbour slot
};

// Phase 2: home operands out of LDS, neighbour operands gathered; the gathers of all kHMulti batches are issued before the first is used.
// slot0 = global slot of home lane 0.
struct NbRegs { u32x4 xy, zm; u32x2 co; };
DEVFN NbRegs nb_issue(const Sorted &so, uint32_t nb) {
    uint32_t off;  // 48 nb as two full-rate instructions (fat_at, pairs.inl)
    asm("v_lshl_add_u32 %0, %1, 1, %1\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(off) : "v"(nb));
    const char *p = reinterpret_cast<const char *>(so.fat) + (size_t)off;
    NbRegs r;
    r.xy = *reinterpret_cast<const u32x4 *>(p); r.zm = *reinterpret_cast<const u32x4 *>(p + 16); r.co = *reinterpret_cast<const u32x2 *>(p + 32);
    return r;
}
DEVFN void exact_one_h(const LdsParams &prm, WaveLdsH &w, BlockLds &bl, const NbRegs &g, uint32_t e, bool active, uint32_t slot0, const EmitTarget &tg,
                       unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t have_res) {
    const uint32_t hl = e >> kHSlotBits, nb = e & ((1u << kHSlotBits) - 1u);
    Fat a, b;
    b.x = __hiloint2double((int)g.xy.y, (int)g.xy.x); b.y = __hiloint2double((int)g.xy.w, (int)g.xy.z); b.z = __hiloint2double((int)g.zm.y, (int)g.zm.x);
    b.pw = g.zm.z; b.res_ord = g.zm.w; b.crm = g.co.x; b.orig = g.co.y;
    const u32x4 axy = w.hxy[hl], azm = w.hzm[hl];
    const u32x2 aco = w.hco[hl];
    a.x = __hiloint2double((int)axy.y, (int)axy.x); a.y = __hiloint2double((int)axy.w, (int)axy.z); a.z = __hiloint2double((int)azm.y, (int)azm.x);
    a.pw = azm.z; a.res_ord = azm.w; a.crm = aco.x; a.orig = aco.y;
    exact_tail(prm, bl, a, b, active, slot0 + hl, nb, tg, result, lane, wflags, have_res);
}
// Entries [first, first + count) of the queue, count <= 64 kHMulti, as up to kHMulti batches whose gathers are issued together.
// (Inactive lanes carry entry 0: home lane 0, slot 0 -- in bounds, ignored.)
DEVFN void exact_batches_h(const LdsParams &prm, WaveLdsH &w, BlockLds &bl, const Sorted &so, uint32_t first, uint32_t count, uint32_t slot0, const EmitTarget &tg,
                           unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t have_res) {
    uint32_t e[kHMulti];
    NbRegs g[kHMulti];
    wave_lds_fence();  // lanes read entries other lanes wrote
#pragma unroll
    for (uint32_t k = 0; k < kHMulti; k++) e[k] = (64u * k + lane < count) ? w.queue[first + 64u * k + lane] : 0u;
    wave_lds_fence();
#pragma unroll
    for (uint32_t k = 0; k < kHMulti; k++)
        if (64u * k < count) g[k] = nb_issue(so, e[k] & ((1u << kHSlotBits) - 1u));   // (wave-uniform condition)
#pragma unroll
    for (uint32_t k = 0; k < kHMulti; k++)
        if (64u * k < count) exact_one_h(prm, w, bl, g[k], e[k], 64u * k + lane < count, slot0, tg, result, lane, wflags, have_res);
}

// WAVES per block: kHWaves for inputs that fill the chip; 4 for small ones, whose few tasks then spread over more CUs
// SPLIT: 1, or 4 = a task's five window kinds are shared out over four waves ({0, 1}, {2}, {3}, {4}), or 8 = two waves per kind set on
// alternate 16-test blocks: a small input has few tasks and each is a long chain of dependent round trips, so more waves on a
// fraction of the chain each is what shortens the launch
template <int WAVES, int SPLIT>
__global__ __launch_bounds__(WAVES * 64, WAVES == kHWaves ? kHWavesPerSimd : 4) void k_pairs_h(DevAtoms in, const GridParams *gp, const DevParams *dprm, const uint32_t *cell_start, Sorted so, EmitTarget tg,
                                                                                ulonglong2 *hole_list, uint32_t *task_ctr, unsigned long long *result) {
    __shared__ LdsParams prm;
    __shared__ WaveLdsH wl[WAVES];
    __shared__ BlockLds bl;
    load_lds_params(prm, dprm, gp);
    if (threadIdx.x == 0) {
        bl.alloc_state = kAllocEmpty | kChunkRecords;  // "exhausted": the first allocation fetches a chunk
        bl.defer_state = kAllocEmpty | kDeferChunk;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, n_heavy = gp->n_heavy, n_tasks = gp->n_tasks * (uint32_t)SPLIT;  // (wave-tasks)
    const uint32_t wflags = (gp->all_both ? kWaveAllBoth : 0u) | ((dprm->flags & ARP_FLAG_CONTACTS_ONLY) ? kWaveContactsOnly : 0u);
    const uint32_t have_res = in.n_res != 0u ? 1u : 0u;
    const double r2m = gp->r2m;
    WaveLdsH &w = wl[wave];
    // task distribution as in k_pairs: block group (b mod 8) = one XCD = one contiguous eighth of the tasks, static first task per wave
    const uint32_t n_groups = min(8u, gridDim.x), group = blockIdx.x % n_groups;
    const uint32_t g_lo = (uint32_t)(((unsigned long long)n_tasks * group) / n_groups), g_hi = (uint32_t)(((unsigned long long)n_tasks * (group + 1u)) / n_groups);
    uint32_t *ctr = task_ctr + (kEmit * 8 + group) * kTaskCtrStride;
    const uint32_t queue_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)w.queue);
    const uint32_t group_waves = ((gridDim.x - group + n_groups - 1u) / n_groups) * WAVES;
    uint32_t t = g_lo + (blockIdx.x / n_groups) * WAVES + wave;
#pragma unroll 1
    while (t < g_hi) {
        constexpr uint32_t kSubs = SPLIT == 8 ? 2u : 1u;  // SPLIT == 8: two waves per kind set, on alternate 16-test blocks
        const uint32_t slot0 = (t / (uint32_t)SPLIT) * 64u, part = (t % (uint32_t)SPLIT) / kSubs, sub = (t % (uint32_t)SPLIT) % kSubs;
        const int k_lo = SPLIT == 1 ? 0 : (part == 0u ? 0 : (int)part + 1), k_hi = SPLIT == 1 ? 5 : (int)part + 2;  // window kinds of this wave-task
        const uint32_t a = slot0 + lane;  // this lane's home slot
        const bool have = a < n_heavy;
        float4 home = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t cx = 0, cy = 0, cz = 0;
        wave_lds_fence();  // the previous task's batches are done with the home records
        {
            u32x4 hxy = {0u, 0u, 0u, 0u}, hzm = {0u, 0u, 0u, 0u};
            u32x2 hco = {0u, 0u};
            if (have) {
                home = so.rec[a];
                const Fat &f = fat_at<false>(so.fat, a);
                const double fx = f.x, fy = f.y, fz = f.z;
                hxy = u32x4{(uint32_t)__double2loint(fx), (uint32_t)__double2hiint(fx), (uint32_t)__double2loint(fy), (uint32_t)__double2hiint(fy)};
                hzm = u32x4{(uint32_t)__double2loint(fz), (uint32_t)__double2hiint(fz), f.pw, f.res_ord};
                hco = u32x2{f.crm, f.orig};
                const uint32_t c = f.cell;
                cx = c % nx; cy = (c / nx) % ny; cz = c / (nx * ny);
            }
            w.hxy[lane] = hxy; w.hzm[lane] = hzm; w.hco[lane] = hco;
        }
        const uint32_t xlo = cx ? cx - 1 : 0, xhi = min(cx + 1, nx - 1);
        // all five slot windows of this lane up front: ten independent loads in flight instead of five round trips
        uint32_t wlo[5] = {0, 0, 0, 0, 0}, whi[5] = {0, 0, 0, 0, 0};
        if (have) {
            wlo[0] = a + 1; whi[0] = cell_start[(cz * ny + cy) * nx + xhi + 1];
#pragma unroll
            for (int k = 1; k < 5; k++) {
                const int dy = (k == 1) ? 1 : (k - 3);
                const uint32_t zz = cz + (k == 1 ? 0u : 1u);
                const int yy = (int)cy + dy;
                if (yy >= 0 && yy < (int)ny && zz < nzt) {
                    const uint32_t r = (zz * ny + (uint32_t)yy) * nx;
                    wlo[k] = cell_start[r + xlo]; whi[k] = cell_start[r + xhi + 1];
                }
            }
        }
        const float3 hm2 = make_float3(-2.0f * home.x, -2.0f * home.y, -2.0f * home.z);
        const float thr = __double2float_ru(r2m - ((double)home.x * home.x + (double)home.y * home.y + (double)home.z * home.z));
        const uint32_t lane_tag = lane << kHSlotBits;
        uint32_t qlen = 0;   // phase-1 survivors waiting in w.queue (wave-uniform); drained at the end of the task
#pragma unroll 1
        for (int k = k_lo; k < k_hi; k++) {
            uint32_t lo = wlo[0], hi = whi[0];
#pragma unroll
            for (int j = 1; j < 5; j++) if (k == j) { lo = wlo[j]; hi = whi[j]; }
            const bool nonempty = lo < hi;
            const uint32_t L = wave_min_u32(nonempty ? lo : 0xFFFFFFFFu), H = wave_max_u32(nonempty ? hi : 0u);
            if (L >= H) continue;
#pragma unroll 1
            for (uint32_t cs = L; cs < H; cs += kHChunk) {
                const uint32_t ce = min(cs + kHChunk, H);
                const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
                const uint32_t len = (nonempty && j1 > j0) ? j1 - j0 : 0u;
                if (!__any(len != 0u)) continue;
                wave_lds_fence();  // previous chunk fully consumed
                for (uint32_t p = cs + lane; p < ce; p += 64u) w.nrec[p - cs] = so.rec[p];
                wave_lds_fence();
                const uint32_t off = len ? j0 - cs : 0u;
#pragma unroll 1
                for (uint32_t it0 = sub * kBlock; __any(it0 < len); it0 += kSubs * kBlock) {
                    const uint32_t wbase = it0 < len ? off + it0 : 0u;
                    const float4 *win = w.nrec + wbase;
                    uint32_t mask = 0;
#pragma unroll
                    for (uint32_t u0 = 0; u0 < kBlock; u0 += kReadAhead) {
                        float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) { const float4 r = win[u0 + u]; rx[u] = r.x; ry[u] = r.y; rz[u] = r.z; rw[u] = r.w; }
                        float acc[kReadAhead];
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rx[u], hm2.x, rw[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(ry[u], hm2.y, acc[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rz[u], hm2.z, acc[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) push_pass(mask, acc[u], thr);
                    }
                    const uint32_t rem = len > it0 ? len - it0 : 0u;  // tests past the window end read other atoms: drop them
                    if (rem < kBlock) mask &= ~((1u << (kBlock - rem)) - 1u);
                    // (mask bit 31 - lz <-> test u = lz - (32 - kBlock) <-> neighbour slot cs + wbase + u)
                    const uint32_t tag = lane_tag + (cs + wbase - (32u - kBlock));
                    while (__any(mask != 0u)) {
                        const uint32_t q0 = __builtin_amdgcn_readfirstlane(qlen);
                        const unsigned long long m = compact_round_x(mask, tag, queue_lds + 4u * q0);
                        uint32_t q1 = q0 + (uint32_t)__popcll(m);
                        if (q1 >= 64u * kHMulti) {
                            q1 -= 64u * kHMulti;
                            exact_batches_h(prm, w, bl, so, q1, 64u * kHMulti, slot0, tg, result, lane, wflags, have_res);
                        }
                        qlen = q1;
                    }
                }
            }
        }
        if (qlen) exact_batches_h(prm, w, bl, so, 0u, qlen, slot0, tg, result, lane, wflags, have_res);  // the home records go with the task: drain
        uint32_t nxt = 0;
        if (lane == 0) nxt = atomicAdd(ctr, 1u);
        t = g_lo + group_waves + __builtin_amdgcn_readfirstlane(nxt);
    }
    emit_epilogue(bl, hole_list + blockIdx.x, tg);
}

// single-pass emit + hole fix-up through k_pairs_h: leaves result[0] = number of pairs, out[0..P) contiguous
void launch_emit_h(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof) {
    EmitTarget tg{out, capacity, ws.scratch, ws.scratch_cap, ws.defer_list, ws.defer_cap};
    const uint32_t tasks = (in.n + 63u) / 64u;
    // few tasks: 4-wave blocks reach more CUs, and every task is shared out over four or eight waves (6bft: 128 tasks)
    // measured on S2 clouds (tests/debug/sweep_split.sh): 8 waves per task win below ~800 tasks, 4 up to ~3000, whole tasks in 12-wave blocks beyond
    const bool small = tasks < 3072u, tiny = tasks < 768u;
    static const int force = [] { const char *e = getenv("ARP_H_SPLIT"); return e ? atoi(e) : 0; }();  // experiments: 1 / 4 / 8 in 4-wave blocks, -1 = the 12-wave kernel
    uint32_t split = tiny ? 8u : (small ? 4u : 1u);
    bool small_blocks = small;
    if (force == 1 || force == 4 || force == 8) { split = (uint32_t)force; small_blocks = true; }
    if (force == -1) { split = 1u; small_blocks = false; }
    const uint32_t per = small_blocks ? 4u : (uint32_t)kHWaves, cap = small_blocks ? 1536u : kHBlocks, want = (split * tasks + per - 1u) / per;
    const uint32_t nb = want < 1 ? 1 : (want > cap ? cap : want);
    if (prof) prof->begin("pairs_emit", st);
#define ARP_LAUNCH_H(W, S) hipLaunchKernelGGL((k_pairs_h<W, S>), dim3(nb), dim3(W * 64), 0, st, in, (const GridParams *)ws.grid, (const DevParams *)ws.params, \
                                              (const uint32_t *)ws.cell_start, ws.sorted, tg, ws.hole_list, ws.task_ctr, ws.result)
    if (small_blocks && split == 8u) ARP_LAUNCH_H(4, 8);
    else if (small_blocks && split == 4u) ARP_LAUNCH_H(4, 4);
    else if (small_blocks) ARP_LAUNCH_H(4, 1);
    else ARP_LAUNCH_H(kHWaves, 1);
#undef ARP_LAUNCH_H
    if (prof) { prof->end(st); prof->begin("pairs_deferred", st); }
    hipLaunchKernelGGL(k_pairs_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg, ws.hole_list + nb, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_fixup", st); }
    hipLaunchKernelGGL(k_fixup, dim3(256), dim3(kFixThreads), 0, st, (const ulonglong2 *)ws.hole_list, nb + kDeferBlocks, (const GridParams *)ws.grid, tg, ws.result);
    if (prof) prof->end(st);
}
static_assert(kHBlocks + 384u <= kMaxHoles && 1536u + 384u <= kMaxHoles, "hole list: one entry per block of either kernel");
