// The single-pass emitter, block-cooperative form.  Included by kernels.hip after pairs_lds.inl, inside namespace arp.
//
// What the two earlier emit kernels measured (profiles/r02_emit_kernels.txt):
//   * k_pairs<kEmit> (pairs.inl) gathers both 48-byte records of every surviving pair from global memory: 3.4 GB of L2->L1 line
//     traffic per launch for 0.5 GB of algorithmic bytes, a plateau of ~240 us at 4, 5, 6 and 8 waves per SIMD -- bound by the
//     scattered-gather path, not by arithmetic (the exact phase went from 145 to 80 vector instructions per batch for 4 %).
//   * k_pairs_x (pairs_lds.inl) stages full records per WAVE and runs the exact phase out of LDS: 6x less L2 traffic, but 10.5 KB
//     of LDS per wave caps it at 3 waves per SIMD, where a wave's own instruction stream (one instruction per ~8 cycles,
//     tests/microbench/valu_rate.hip) sets the pace: 295 us.
// Here FOUR waves (256 consecutive home slots) share one staged chunk -- neighbouring tasks see almost the same slot windows, so the
// chunk is staged once per block (1.11 records per home slot and window kind instead of 1.64) -- and the home atoms' exact records
// stay in the owning lane's registers, fetched per pair by ds_bpermute.  That is 5 KB of LDS per wave: 5 waves per SIMD.
constexpr int kBWaves = 4;                   // waves per block = wave-tasks per block-task
constexpr uint32_t kBChunk = 320;            // staged neighbour records per chunk (a block's 256 home slots + ~4 cells of halo)
constexpr uint32_t kBBlocksPerCu = 5;
constexpr uint32_t kBBlocks = 256 * kBBlocksPerCu;

struct BlockLdsB {
    f32x4 rec[kBChunk + kBlock];             // f32 prefilter records of the staged chunk (+ kBlock: over-reads stay in bounds)
    u32x4 xy[kBChunk];                       // Fat part 0: {x, y} as raw words
    u32x4 zm[kBChunk];                       // Fat part 1: {z, pw, res_ord}
    u32x2 co[kBChunk];                       // Fat part 2, first half: {crm, orig}
    uint32_t queue[kBWaves][kXQueue];        // per wave: phase-1 survivors, home lane << 16 | record index in the chunk
    uint32_t wave_lo[kBWaves][5], wave_hi[kBWaves][5];   // per wave and window kind: union of the lanes' windows
    uint32_t next_task;
};

// Phase 2: the home operand comes out of the owning lane's registers (ds_bpermute: the LDS crossbar, no bank conflicts, no LDS space),
// the neighbour operand out of the staged chunk.
struct HomeExact { uint32_t xlo, xhi, ylo, yhi, zlo, zhi, pw, res_ord, crm, orig; };
DEVFN void exact_batch_b(const LdsParams &prm, BlockLdsB &sb, BlockLds &bl, const HomeExact &h, uint32_t e, bool active, uint32_t slot0, uint32_t cs,
                         const EmitTarget &tg, unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t have_res) {
#if defined(ARP_ABLATE) && ARP_ABLATE == 21   // timing ablation: no exact phase at all
    return;
#endif
    const uint32_t hl = e >> 16, bi = e & 0xFFFFu;
    const int src = (int)(hl << 2);
    Fat a, b;
    {
        auto pull = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v); };
        const uint32_t axl = pull(h.xlo), axh = pull(h.xhi), ayl = pull(h.ylo), ayh = pull(h.yhi), azl = pull(h.zlo), azh = pull(h.zhi);
        a.pw = pull(h.pw); a.res_ord = pull(h.res_ord); a.crm = pull(h.crm); a.orig = pull(h.orig);
        a.x = __hiloint2double((int)axh, (int)axl); a.y = __hiloint2double((int)ayh, (int)ayl); a.z = __hiloint2double((int)azh, (int)azl);
        const u32x4 bxy = sb.xy[bi], bzm = sb.zm[bi];
        const u32x2 bco = sb.co[bi];
        b.x = __hiloint2double((int)bxy.y, (int)bxy.x); b.y = __hiloint2double((int)bxy.w, (int)bxy.z); b.z = __hiloint2double((int)bzm.y, (int)bzm.x);
        b.pw = bzm.z; b.res_ord = bzm.w; b.crm = bco.x; b.orig = bco.y;
    }
#if defined(ARP_ABLATE) && ARP_ABLATE == 24   // timing ablation: operand fetch only
    if (a.x + b.x + a.y + b.y + a.z + b.z == 1.2345e300 && a.pw + b.pw + a.res_ord + b.res_ord + a.crm + b.crm + a.orig + b.orig == 77u) atomicOr(&result[1], 32ull);
    return;
#endif
    exact_tail(prm, bl, a, b, active, slot0 + hl, cs + bi, tg, result, lane, wflags, have_res);
}

__global__ __launch_bounds__(kBWaves * 64, kBBlocksPerCu) void k_pairs_b(DevAtoms in, const GridParams *gp, const DevParams *dprm, const uint32_t *cell_start, Sorted so,
                                                                         EmitTarget tg, ulonglong2 *hole_list, uint32_t *task_ctr, unsigned long long *result) {
    __shared__ LdsParams prm;
    __shared__ BlockLdsB sb;
    __shared__ BlockLds bl;
    load_lds_params(prm, dprm, gp);
    if (threadIdx.x == 0) {
        bl.alloc_state = kAllocEmpty | kChunkRecords;  // "exhausted": the first allocation fetches a chunk
        bl.defer_state = kAllocEmpty | kDeferChunk;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, n_heavy = gp->n_heavy, n_tasks = gp->n_tasks;
    const uint32_t wflags = (gp->all_both ? kWaveAllBoth : 0u) | ((dprm->flags & ARP_FLAG_CONTACTS_ONLY) ? kWaveContactsOnly : 0u);
    const uint32_t have_res = in.n_res != 0u ? 1u : 0u;
    const double r2m = gp->r2m;
    // Block-tasks (kBWaves consecutive wave-tasks), handed out like the wave-tasks of k_pairs: block group (b mod 8) shares an XCD and
    // owns one contiguous eighth of the range; the first block-task of a block is static, later ones come from the group's counter.
    const uint32_t n_btasks = (n_tasks + kBWaves - 1u) / kBWaves;
    const uint32_t n_groups = min(8u, gridDim.x), group = blockIdx.x % n_groups;
    const uint32_t g_lo = (uint32_t)(((unsigned long long)n_btasks * group) / n_groups), g_hi = (uint32_t)(((unsigned long long)n_btasks * (group + 1u)) / n_groups);
    uint32_t *ctr = task_ctr + (kEmit * 8 + group) * kTaskCtrStride;
    const uint32_t group_blocks = (gridDim.x - group + n_groups - 1u) / n_groups;
    const uint32_t queue_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)sb.queue[wave]);
    const f32x4 *grec = reinterpret_cast<const f32x4 *>(so.rec);
    const u32x4 *gfat4 = reinterpret_cast<const u32x4 *>(so.fat);
    const u32x2 *gfat2 = reinterpret_cast<const u32x2 *>(so.fat);
    const uint32_t lane_tag = lane << 16;
    uint32_t bt = g_lo + blockIdx.x / n_groups;
    uint32_t abl_acc = 0; (void)abl_acc;
#pragma unroll 1
    while (bt < g_hi) {
        // ---- this wave's task: home records (kept in registers for the whole block-task) and slot windows
        const uint32_t t = bt * kBWaves + wave, slot0 = t * 64u, a = slot0 + lane;
        const bool have = t < n_tasks && a < n_heavy;
        const uint32_t ha = have ? a : (n_heavy ? n_heavy - 1u : 0u);
        const f32x4 home = grec[ha];
        const u32x4 p0 = gfat4[3u * ha], p1 = gfat4[3u * ha + 1u], p2 = gfat4[3u * ha + 2u];  // {x, y} {z, pw, res_ord} {crm, orig, cell, attr}
        const HomeExact hx{p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y};
        Five wlo, whi;
        windows_issue(wlo, whi, cell_start, a, p2.z, have, nx, ny, nzt);
        // per-lane constants of the prefilter: -2 h (exact in f32) and the threshold r2m - |h|^2, rounded up
        const float3 hm2 = make_float3(-2.0f * home.x, -2.0f * home.y, -2.0f * home.z);
        const float thr = __double2float_ru(r2m - ((double)home.x * home.x + (double)home.y * home.y + (double)home.z * home.z));
        {
            auto lo_of = [](uint32_t lo, uint32_t hi) { return lo < hi ? lo : 0xFFFFFFFFu; };
            auto hi_of = [](uint32_t lo, uint32_t hi) { return lo < hi ? hi : 0u; };
            const Five L{wave_min_u32(lo_of(wlo.v0, whi.v0)), wave_min_u32(lo_of(wlo.v1, whi.v1)), wave_min_u32(lo_of(wlo.v2, whi.v2)),
                         wave_min_u32(lo_of(wlo.v3, whi.v3)), wave_min_u32(lo_of(wlo.v4, whi.v4))};
            const Five H{wave_max_u32(hi_of(wlo.v0, whi.v0)), wave_max_u32(hi_of(wlo.v1, whi.v1)), wave_max_u32(hi_of(wlo.v2, whi.v2)),
                         wave_max_u32(hi_of(wlo.v3, whi.v3)), wave_max_u32(hi_of(wlo.v4, whi.v4))};
            if (lane < 5u) { sb.wave_lo[wave][lane] = sel5((int)lane, L); sb.wave_hi[wave][lane] = sel5((int)lane, H); }
        }
        if (threadIdx.x == 0) sb.next_task = atomicAdd(ctr, 1u);  // the block's next task, drawn early: its latency hides behind this one
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < 5; k++) {
            // union of the kind over the block (every thread reads the same eight words: block-uniform control flow, no vote needed)
            uint32_t L = 0xFFFFFFFFu, H = 0u;
#pragma unroll
            for (int w = 0; w < kBWaves; w++) { L = min(L, sb.wave_lo[w][k]); H = max(H, sb.wave_hi[w][k]); }
            L = (uint32_t)__builtin_amdgcn_readfirstlane((int)L); H = (uint32_t)__builtin_amdgcn_readfirstlane((int)H);
            if (L >= H) continue;
            const uint32_t lo = sel5(k, wlo), hi = sel5(k, whi);
#pragma unroll 1
            for (uint32_t cs = L; cs < H; cs += kBChunk) {
                const uint32_t ce = min(cs + kBChunk, H);
                {   // skip a chunk no wave's window range meets (only unions far larger than a block's own span have such chunks)
                    bool any = false;
#pragma unroll
                    for (int w = 0; w < kBWaves; w++) any = any || (sb.wave_lo[w][k] < sb.wave_hi[w][k] && max(sb.wave_lo[w][k], cs) < min(sb.wave_hi[w][k], ce));
                    if (!__builtin_amdgcn_readfirstlane((int)any)) continue;
                }
                // stage the chunk's full records, coalesced: 16 + 40 bytes per slot, once per block
                for (uint32_t q = threadIdx.x; q < ce - cs; q += kBWaves * 64u) {
                    const uint32_t p = cs + q;
                    const f32x4 rr = grec[p];
                    const u32x4 q0 = gfat4[3u * p], q1 = gfat4[3u * p + 1u];
                    const u32x2 q2 = gfat2[6u * p + 4u];
                    sb.rec[q] = rr; sb.xy[q] = q0; sb.zm[q] = q1; sb.co[q] = q2;
                }
                __syncthreads();
                const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
                const uint32_t len = (lo < hi && j1 > j0) ? j1 - j0 : 0u;
                const uint32_t off = len ? j0 - cs : 0u;
                uint32_t qlen = 0;  // survivors waiting in this wave's queue (wave-uniform); they refer to THIS chunk
#if defined(ARP_ABLATE) && ARP_ABLATE == 23   // timing ablation: staging and barriers only
                if (len == 0xFFFFFFFFu)
#endif
#pragma unroll 1
                for (uint32_t it0 = 0; __any(it0 < len); it0 += kBlock) {
                    // Phase 1: kBlock prefilter tests per lane, results pushed into a per-lane bit mask (test u -> bit kBlock-1-u).
                    // Lanes whose window is exhausted read records 0..kBlock-1 (any staged data will do: their bits are dropped).
                    const uint32_t wbase = it0 < len ? off + it0 : 0u;
                    const f32x4 *win = sb.rec + wbase;
                    uint32_t mask = 0;
#pragma unroll
                    for (uint32_t u0 = 0; u0 < kBlock; u0 += kReadAhead) {
                        float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) { const f32x4 r = win[u0 + u]; rx[u] = r.x; ry[u] = r.y; rz[u] = r.z; rw[u] = r.w; }
                        // |n|^2 - 2 n.h against thr = r2m - |h|^2: 5 VALU per test.  The three FMAs of ONE test are a dependent chain (~2 ns per
                        // link on a SIMD, tests/microbench/valu_rate.hip); issued test-major they serialise, so the loop runs link-major over the
                        // kReadAhead tests in flight: neighbouring instructions are independent.
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
                    // Compaction: one round per surviving test of the busiest lane; every round appends <= 64 entries
                    const uint32_t tag = lane_tag + wbase - (32u - kBlock);
#if defined(ARP_ABLATE) && ARP_ABLATE == 22   // timing ablation: prefilter only (no compaction, no exact phase)
                    abl_acc += (uint32_t)__popc(mask); mask = 0u;
#endif
                    while (__any(mask != 0u)) {
                        const uint32_t q0 = __builtin_amdgcn_readfirstlane(qlen);
                        const unsigned long long m = compact_round_x(mask, tag, queue_lds + 4u * q0);
                        uint32_t q1 = q0 + (uint32_t)__popcll(m);
                        if (q1 >= 64) {
                            q1 -= 64;
                            wave_lds_fence();  // lanes read entries other lanes wrote
                            const uint32_t e = sb.queue[wave][q1 + lane];
                            wave_lds_fence();
                            exact_batch_b(prm, sb, bl, hx, e, true, slot0, cs, tg, result, lane, wflags, have_res);
                        }
                        qlen = q1;
                    }
                }
                if (qlen) {  // the survivors left at the chunk end: their records are about to be overwritten
                    const bool act = lane < qlen;
                    wave_lds_fence();
                    const uint32_t e = act ? sb.queue[wave][lane] : 0u;
                    wave_lds_fence();
                    exact_batch_b(prm, sb, bl, hx, e, act, slot0, cs, tg, result, lane, wflags, have_res);
                }
                __syncthreads();  // every wave is done with the chunk before the next one is staged
            }
        }
        bt = g_lo + group_blocks + sb.next_task;
        __syncthreads();  // (next_task and the per-wave unions are rewritten by the next block-task)
    }
#if defined(ARP_ABLATE) && ARP_ABLATE == 22
    if (abl_acc == 0xFFFFFFFFu) atomicOr(&result[1], 32ull);
#endif
    emit_epilogue(bl, hole_list + blockIdx.x, tg);
}

// single-pass emit + hole fix-up through k_pairs_b: leaves result[0] = number of pairs, out[0..P) contiguous
void launch_emit_b(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof) {
    EmitTarget tg{out, capacity, ws.scratch, ws.scratch_cap, ws.defer_list, ws.defer_cap};
    const uint32_t tasks = (in.n + 63u) / 64u, want = (tasks + kBWaves - 1) / kBWaves;
    const uint32_t nb = want < 1 ? 1 : (want > kBBlocks ? kBBlocks : want);
    if (prof) prof->begin("pairs_emit", st);
    hipLaunchKernelGGL(k_pairs_b, dim3(nb), dim3(kBWaves * 64), 0, st, in, (const GridParams *)ws.grid, (const DevParams *)ws.params, (const uint32_t *)ws.cell_start,
                       ws.sorted, tg, ws.hole_list, ws.task_ctr, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_deferred", st); }
    hipLaunchKernelGGL(k_pairs_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg, ws.hole_list + nb, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_fixup", st); }
    hipLaunchKernelGGL(k_fixup, dim3(256), dim3(kFixThreads), 0, st, (const ulonglong2 *)ws.hole_list, nb + kDeferBlocks, (const GridParams *)ws.grid, tg, ws.result);
    if (prof) prof->end(st);
}
