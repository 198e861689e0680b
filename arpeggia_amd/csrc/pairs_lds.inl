// The single-pass emitter with an LDS-resident exact phase.  Included by kernels.hip after pairs.inl, inside namespace arp.
//
// k_pairs<kEmit> (pairs.inl) fetched both 48-byte records of every surviving pair with six scattered global loads per lane
// and batch; measured (profiles/r02_*), that gather chain -- not the arithmetic -- set the pace: cutting the exact phase from
// 145 to 80 vector instructions per batch moved the kernel by 4 %.  Here the wave stages the FULL records of each slot
// window once, coalesced, next to the f32 prefilter records, and keeps the exact records of its 64 home atoms in LDS for
// the whole task; phase 2 then reads both operands of a pair with ds_read only.  Steady state has no vector-memory load
// behind which a record store must be acknowledged (gfx9 counts loads and stores in one in-order vmcnt).
//
// Price: 10.3 KB of LDS per wave (12 waves per CU in 4-wave blocks), and survivors cannot outlive the staged chunk, so the
// queue is drained (one partial batch) at every chunk end.
#ifdef ARP_STAMP   // diagnostic build: s_memtime stamps around the segments of a wave's task loop, summed into result[8..]
#define STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime(); const unsigned long long st_begin = st_t;
#define STAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_t; st_t = now_; }
#define STAMP_COUNT(k) { st_acc[k] += 1ull; }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_COUNT(k)
#endif
// Native vector types for everything that is held in registers between a load and an LDS store: HIP's uint4 / float4 are structs
// whose copies become memcpy calls, and a struct of them that lives across a loop stays in scratch memory.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int kXWaves = 4;                   // waves per block
constexpr uint32_t kXChunk = 128;            // staged neighbour records per chunk
constexpr uint32_t kXBlocks = 768;           // 3 blocks of 4 waves per CU
constexpr uint32_t kXQueue = 128;
constexpr uint32_t kXReadAhead = 8;           // LDS reads in flight per lane in the prefilter (registers are not the limit here: LDS caps the waves)

struct WaveLdsX {
    f32x4 rec[kXChunk + kBlock];             // f32 prefilter records of the staged chunk (+ kBlock: over-reads stay in bounds)
    u32x4 xy[kXChunk];                       // Fat part 0: {x, y} as raw words
    u32x4 zm[kXChunk];                       // Fat part 1: {z, pw, res_ord}
    u32x2 co[kXChunk];                       // Fat part 2, first half: {crm, orig}
    u32x4 hxy[64]; u32x4 hzm[64]; u32x2 hco[64];     // the same three parts of the task's 64 home atoms
    uint32_t queue[kXQueue];                 // phase-1 survivors: home lane << 16 | record index in the chunk
};

// compact_round (pairs.inl) with 4-byte entries: every lane with a surviving test appends tag + (31 - index of its highest
// set bit) at byte address qaddr + 4 * (its rank among those lanes) and clears the bit.  tag = home lane << 16 | (chunk index of
// test 0 of the block - (32 - kBlock)).
DEVFN unsigned long long compact_round_x(uint32_t &mask, uint32_t tag, uint32_t qaddr) {
    unsigned long long m, save;
    uint32_t t, lz, ent, bm;
    asm volatile(
        "v_cmp_ne_u32 vcc, 0, %[mask]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_ffbh_u32 %[lz], %[mask]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[qaddr]\n\t"
        "v_add_u32 %[ent], %[tag], %[lz]\n\t"
        "v_lshrrev_b32 %[bm], %[lz], %[top]\n\t"
        "s_mov_b64 %[m], vcc\n\t"
        "s_and_saveexec_b64 %[save], vcc\n\t"
        "ds_write_b32 %[t], %[ent]\n\t"
        "v_xor_b32 %[mask], %[mask], %[bm]\n\t"
        "s_mov_b64 exec, %[save]"
        : [mask] "+v"(mask), [t] "=&v"(t), [lz] "=&v"(lz), [ent] "=&v"(ent), [bm] "=&v"(bm), [m] "=&s"(m), [save] "=&s"(save)
        : [tag] "v"(tag), [qaddr] "s"(qaddr), [top] "s"(0x80000000u)
        : "vcc", "memory");
    return m;
}

// The part of phase 2 behind the operand fetch: exact f64 decision, pair filter, classification, compacted coalesced store.
// slot_a / slot_b: global slots of the two records (for the deferred list, whose entries are global slot pairs).
DEVFN void exact_tail(const LdsParams &prm, BlockLds &bl, const Fat &a, const Fat &b, bool active, uint32_t slot_a, uint32_t slot_b, const EmitTarget &tg,
                      unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t have_res) {
    const double s = sq_dist(a.x, a.y, a.z, b.x, b.y, b.z);
    const int o = (wflags & kWaveAllBoth) ? orient_all_both(a, b) : orient(a, b);
    bool valid = active & (s <= prm.r2) & (o != 0);  // rstar: inclusive
    const bool swap = o == 2;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (valid) {
        r.w = classify_fast(prm, s, a.pw, b.pw, have_res);
        r.x = swap ? b.orig : a.orig; r.y = swap ? a.orig : b.orig;
        r.z = __float_as_uint(dist_f32(s));
    }
    if (wflags & kWaveContactsOnly) valid = valid && r.w != 0u;  // ARP_FLAG_CONTACTS_ONLY (kDeferKind != 0 stays)
    {   // candidates whose rules need a probe go to the deferred pass (k_pairs_deferred), as global slot pairs
        const bool defer = valid && r.w == kDeferKind;
        const unsigned long long dm = __ballot(defer);
        if (dm) {
            const Slots ds = alloc_chunked<kDeferChunk>(bl.defer_state, &result[3], (uint32_t)__popcll(dm), lane);
            if (defer) {
                const uint32_t dr = mbcnt(dm);
                const unsigned long long p = dr < ds.n0 ? ds.pos0 + dr : ds.pos1 + (dr - ds.n0);
                if (p < tg.defer_cap) tg.defer_list[p] = make_uint2(slot_a, slot_b); else atomicOr(&result[1], 8ull);
            }
            valid = valid && !defer;
        }
    }
    const unsigned long long vm = __ballot(valid);
    const uint32_t n = (uint32_t)__popcll(vm);
    if (n) {  // compacted, coalesced store of the batch's records straight from registers
        const Slots sl = alloc_chunked<kChunkRecords>(bl.alloc_state, &result[2], n, lane);
        const uint32_t rank = mbcnt(vm);
        if (sl.n0 == n && sl.pos0 + n <= tg.capacity) {  // one run inside the caller's buffer: scalar base + 32-bit lane offset
            uint4 *run = reinterpret_cast<uint4 *>(tg.out) + sl.pos0;
            if (valid) store_record(&run[rank], r);
        } else if (valid) {  // the run crosses a chunk end or the end of the caller's buffer (scratch until k_fixup)
            uint4 *d = emit_slot(tg, rank < sl.n0 ? sl.pos0 + rank : sl.pos1 + (rank - sl.n0), result);
            if (d) store_record(d, r);
        }
    }
}

// Phase 2 on up to 64 survivors, both operands out of LDS.  slot0 / cs: global slot of home lane 0 / of chunk record 0 (for
// the deferred list, whose entries are global slot pairs).
DEVFN void exact_batch_x(const LdsParams &prm, WaveLdsX &w, BlockLds &bl, uint32_t e, bool active, uint32_t slot0, uint32_t cs,
                         const EmitTarget &tg, unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t have_res) {
    const uint32_t hl = e >> 16, bi = e & 0xFFFFu;
    Fat a, b;
    {   // (inactive lanes of a partial batch read record 0 of both tables: in bounds, ignored)
        const u32x4 axy = w.hxy[hl], bxy = w.xy[bi];
        const u32x4 azm = w.hzm[hl], bzm = w.zm[bi];
        const u32x2 aco = w.hco[hl], bco = w.co[bi];
        a.x = __hiloint2double((int)axy.y, (int)axy.x); a.y = __hiloint2double((int)axy.w, (int)axy.z); a.z = __hiloint2double((int)azm.y, (int)azm.x);
        a.pw = azm.z; a.res_ord = azm.w; a.crm = aco.x; a.orig = aco.y;
        b.x = __hiloint2double((int)bxy.y, (int)bxy.x); b.y = __hiloint2double((int)bxy.w, (int)bxy.z); b.z = __hiloint2double((int)bzm.y, (int)bzm.x);
        b.pw = bzm.z; b.res_ord = bzm.w; b.crm = bco.x; b.orig = bco.y;
    }
    exact_tail(prm, bl, a, b, active, slot0 + hl, cs + bi, tg, result, lane, wflags, have_res);
}

// Everything a wave fetches from global memory is requested one step ahead of its use: the records of chunk i+1 while chunk i is
// searched, the home records and slot windows of the next task during the current one, the task after that from the group counter.
// With 3 waves per SIMD an exposed round trip (1-2 us under load) is not covered by other waves.
// Five values selected by a (wave-uniform) kind, passed BY VALUE everywhere.  Deliberately neither an array nor reached through
// references or capturing lambdas: the optimiser rewrites select chains over memory into indexed loads, and the five-element
// tables then live in scratch memory -- vector-memory round trips in the inner loops.
struct Five { uint32_t v0, v1, v2, v3, v4; };
DEVFN uint32_t sel5(int k, Five f) {
    uint32_t r = f.v0;
    r = k == 1 ? f.v1 : r; r = k == 2 ? f.v2 : r; r = k == 3 ? f.v3 : r; r = k == 4 ? f.v4 : r;
    return r;
}
// chunk iterator: from (kind k, chunk start cs) to the next chunk that some lane's window of its kind meets; k == 5 when done
DEVFN void next_chunk(int &k, uint32_t &cs, bool first, Five wlo, Five whi, Five Lk, Five Hk, uint32_t chunk) {
    uint32_t L, H;
    if (first) { k = 0; L = Lk.v0; H = Hk.v0; cs = L; } else { L = sel5(k, Lk); H = sel5(k, Hk); cs += chunk; }
    for (;;) {
        if (cs >= H || L >= H) {
            if (++k >= 5) return;
            L = sel5(k, Lk); H = sel5(k, Hk); cs = L;
            continue;
        }
        const uint32_t lo = sel5(k, wlo), hi = sel5(k, whi), ce = min(cs + chunk, H);
        if (__any(lo < hi && max(lo, cs) < min(hi, ce))) return;
        cs += chunk;
    }
}
struct StageRegs {  // one chunk in flight: two slots per lane
    f32x4 ra, rb; u32x4 p0a, p0b, p1a, p1b; u32x2 p2a, p2b;
};
DEVFN void stage_issue(StageRegs &sr, const Sorted &so, const Fat *fat, uint32_t cs, uint32_t ce, uint32_t lane) {
    // unconditional loads from a clamped slot: predicated loads serialise
    const uint32_t pa = min(cs + lane, ce - 1u), pb = min(cs + lane + 64u, ce - 1u);
    const f32x4 *rec = reinterpret_cast<const f32x4 *>(so.rec);
    const u32x4 *fat4 = reinterpret_cast<const u32x4 *>(fat);
    const u32x2 *fat2 = reinterpret_cast<const u32x2 *>(fat);
    sr.ra = rec[pa]; sr.rb = rec[pb];
    sr.p0a = fat4[3u * pa]; sr.p0b = fat4[3u * pb];
    sr.p1a = fat4[3u * pa + 1u]; sr.p1b = fat4[3u * pb + 1u];
    sr.p2a = fat2[6u * pa + 4u]; sr.p2b = fat2[6u * pb + 4u];
}
DEVFN void stage_commit(const StageRegs &sr, WaveLdsX &w, uint32_t cs, uint32_t ce, uint32_t lane) {
    if (cs + lane < ce) { w.rec[lane] = sr.ra; w.xy[lane] = sr.p0a; w.zm[lane] = sr.p1a; w.co[lane] = sr.p2a; }
    if (cs + lane + 64u < ce) { w.rec[lane + 64u] = sr.rb; w.xy[lane + 64u] = sr.p0b; w.zm[lane + 64u] = sr.p1b; w.co[lane + 64u] = sr.p2b; }
}
struct HomeRegs {  // a task's home atoms as loaded: prefilter record + the three parts of the exact record
    f32x4 rec; u32x4 p0, p1, p2;
};
DEVFN void home_issue(HomeRegs &h, const Sorted &so, const Fat *fat, uint32_t a, uint32_t n_heavy) {
    const uint32_t p = min(a, n_heavy ? n_heavy - 1u : 0u);
    const u32x4 *fat4 = reinterpret_cast<const u32x4 *>(fat);
    h.rec = reinterpret_cast<const f32x4 *>(so.rec)[p];
    h.p0 = fat4[3u * p]; h.p1 = fat4[3u * p + 1u]; h.p2 = fat4[3u * p + 2u];  // {crm, orig, cell, attr}
}
// the five slot windows of a home atom in cell c (half shell: rest of the home row, row y+1, the three rows of layer z+1)
DEVFN void windows_issue(Five &wlo, Five &whi, const uint32_t *cell_start, uint32_t a, uint32_t c, bool have, uint32_t nx, uint32_t ny, uint32_t nzt) {
    const uint32_t cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    const uint32_t xlo = cx ? cx - 1 : 0, xhi = min(cx + 1, nx - 1);
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
    if (have) {
        l0 = a + 1; h0 = cell_start[(cz * ny + cy) * nx + xhi + 1];
        const bool yp = cy + 1u < ny, ym = cy > 0u, zp = cz + 1u < nzt;
        const uint32_t r1 = (cz * ny + cy + 1u) * nx, r2 = ((cz + 1u) * ny + cy - 1u) * nx, r3 = ((cz + 1u) * ny + cy) * nx, r4 = ((cz + 1u) * ny + cy + 1u) * nx;
        if (yp) { l1 = cell_start[r1 + xlo]; h1 = cell_start[r1 + xhi + 1]; }
        if (zp && ym) { l2 = cell_start[r2 + xlo]; h2 = cell_start[r2 + xhi + 1]; }
        if (zp) { l3 = cell_start[r3 + xlo]; h3 = cell_start[r3 + xhi + 1]; }
        if (zp && yp) { l4 = cell_start[r4 + xlo]; h4 = cell_start[r4 + xhi + 1]; }
    }
    wlo = Five{l0, l1, l2, l3, l4}; whi = Five{h0, h1, h2, h3, h4};
}

__global__ __launch_bounds__(kXWaves * 64, 3) void k_pairs_x(DevAtoms in, const GridParams *gp, const DevParams *dprm, const uint32_t *cell_start, Sorted so,
                                                              EmitTarget tg, ulonglong2 *hole_list, uint32_t *task_ctr, unsigned long long *result) {
    __shared__ LdsParams prm;
    __shared__ WaveLdsX wl[kXWaves];
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
    WaveLdsX &w = wl[wave];
    // task distribution as in k_pairs: XCD-grouped contiguous ranges, static first task, later ones from the group's counter --
    // drawn one task AHEAD, so that the next task's records can be requested while this one is searched
    const uint32_t n_groups = min(8u, gridDim.x), group = blockIdx.x % n_groups;
    const uint32_t g_lo = (uint32_t)(((unsigned long long)n_tasks * group) / n_groups), g_hi = (uint32_t)(((unsigned long long)n_tasks * (group + 1u)) / n_groups);
    uint32_t *ctr = task_ctr + (kEmit * 8 + group) * kTaskCtrStride;
    const uint32_t queue_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)w.queue);
    const uint32_t group_waves = ((gridDim.x - group + n_groups - 1u) / n_groups) * kXWaves;
    const Fat *fat4 = so.fat;
    const uint32_t lane_tag = lane << 16;
    STAMP_DECL
    uint32_t t = g_lo + (blockIdx.x / n_groups) * kXWaves + wave, t_next;
    {
        uint32_t nxt = 0;
        if (lane == 0) nxt = atomicAdd(ctr, 1u);
        t_next = g_lo + group_waves + __builtin_amdgcn_readfirstlane(nxt);
    }
    HomeRegs hr{};
    Five wlo{0, 0, 0, 0, 0}, whi{0, 0, 0, 0, 0};
    if (t < g_hi) {
        home_issue(hr, so, fat4, t * 64u + lane, n_heavy);
        windows_issue(wlo, whi, cell_start, t * 64u + lane, hr.p2.z, t * 64u + lane < n_heavy, nx, ny, nzt);
    }
#pragma unroll 1
    while (t < g_hi) {
        STAMP(7) STAMP_COUNT(11)
        const uint32_t slot0 = t * 64u;
        const f32x4 home = hr.rec;
        wave_lds_fence();  // the previous task's batches have read their home records
        w.hxy[lane] = hr.p0; w.hzm[lane] = hr.p1; w.hco[lane] = u32x2{hr.p2.x, hr.p2.y};
        // the next task: its home records now, its windows once those have landed (below), the task after it from the counter
        const bool more = t_next < g_hi;
        HomeRegs hn{};
        Five nlo{0, 0, 0, 0, 0}, nhi{0, 0, 0, 0, 0};
        uint32_t grab = 0;
        bool next_windows = false;
        if (more) {
            home_issue(hn, so, fat4, t_next * 64u + lane, n_heavy);
            if (lane == 0) grab = atomicAdd(ctr, 1u);
        }
        // per-lane constants of the prefilter: -2 h (exact in f32) and the threshold r2m - |h|^2, rounded up
        const float3 hm2 = make_float3(-2.0f * home.x, -2.0f * home.y, -2.0f * home.z);
        const float thr = __double2float_ru(r2m - ((double)home.x * home.x + (double)home.y * home.y + (double)home.z * home.z));
        // union [L, H) of every window kind over the wave
        Five Lk, Hk;
        {
            auto lo_of = [](uint32_t lo, uint32_t hi) { return lo < hi ? lo : 0xFFFFFFFFu; };
            auto hi_of = [](uint32_t lo, uint32_t hi) { return lo < hi ? hi : 0u; };
            Lk = Five{wave_min_u32(lo_of(wlo.v0, whi.v0)), wave_min_u32(lo_of(wlo.v1, whi.v1)), wave_min_u32(lo_of(wlo.v2, whi.v2)),
                      wave_min_u32(lo_of(wlo.v3, whi.v3)), wave_min_u32(lo_of(wlo.v4, whi.v4))};
            Hk = Five{wave_max_u32(hi_of(wlo.v0, whi.v0)), wave_max_u32(hi_of(wlo.v1, whi.v1)), wave_max_u32(hi_of(wlo.v2, whi.v2)),
                      wave_max_u32(hi_of(wlo.v3, whi.v3)), wave_max_u32(hi_of(wlo.v4, whi.v4))};
        }
        STAMP(1)
        int k = 0;
        uint32_t cs = 0;
        next_chunk(k, cs, true, wlo, whi, Lk, Hk, kXChunk);
        StageRegs sr{};
        if (k < 5) stage_issue(sr, so, fat4, cs, min(cs + kXChunk, sel5(k, Hk)), lane);
#pragma unroll 1
        while (k < 5) {
            const uint32_t lo = sel5(k, wlo), hi = sel5(k, whi);
            const uint32_t ce = min(cs + kXChunk, sel5(k, Hk));
            const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
            const uint32_t len = (lo < hi && j1 > j0) ? j1 - j0 : 0u;
            wave_lds_fence();  // previous chunk fully consumed (its queue was drained)
            stage_commit(sr, w, cs, ce, lane);
            wave_lds_fence();
            int k2 = k;
            uint32_t cs2 = cs;
            next_chunk(k2, cs2, false, wlo, whi, Lk, Hk, kXChunk);
            if (k2 < 5) stage_issue(sr, so, fat4, cs2, min(cs2 + kXChunk, sel5(k2, Hk)), lane);
            if (more && !next_windows && k >= 1) {  // the next task's home records were requested a whole chunk ago
                next_windows = true;
                windows_issue(nlo, nhi, cell_start, t_next * 64u + lane, hn.p2.z, t_next * 64u + lane < n_heavy, nx, ny, nzt);
            }
            STAMP(2) STAMP_COUNT(10)
            const uint32_t off = len ? j0 - cs : 0u;
            uint32_t qlen = 0;  // survivors waiting in w.queue (wave-uniform); they refer to THIS chunk
#pragma unroll 1
            for (uint32_t it0 = 0; __any(it0 < len); it0 += kBlock) {
                // Phase 1: kBlock prefilter tests per lane, results pushed into a per-lane bit mask (test u -> bit kBlock-1-u).
                // Lanes whose window is exhausted read records 0..kBlock-1 (any staged data will do: their bits are dropped).
                const uint32_t wbase = it0 < len ? off + it0 : 0u;
                const f32x4 *win = w.rec + wbase;
                uint32_t mask = 0;
#pragma unroll
                for (uint32_t u0 = 0; u0 < kBlock; u0 += kXReadAhead) {
                    float rx[kXReadAhead], ry[kXReadAhead], rz[kXReadAhead], rw[kXReadAhead];
#pragma unroll
                    for (uint32_t u = 0; u < kXReadAhead; ++u) { const f32x4 r = win[u0 + u]; rx[u] = r.x; ry[u] = r.y; rz[u] = r.z; rw[u] = r.w; }
                    // |n|^2 - 2 n.h against thr = r2m - |h|^2: 5 VALU per test.  The three FMAs of ONE test are a dependent chain (~2 ns per
                    // link on a SIMD, tests/microbench/valu_rate.hip); issued test-major they serialise, so the loop runs link-major over the
                    // kXReadAhead tests in flight: neighbouring instructions are independent.
                    float acc[kXReadAhead];
#pragma unroll
                    for (uint32_t u = 0; u < kXReadAhead; ++u) acc[u] = __fmaf_rn(rx[u], hm2.x, rw[u]);
#pragma unroll
                    for (uint32_t u = 0; u < kXReadAhead; ++u) acc[u] = __fmaf_rn(ry[u], hm2.y, acc[u]);
#pragma unroll
                    for (uint32_t u = 0; u < kXReadAhead; ++u) acc[u] = __fmaf_rn(rz[u], hm2.z, acc[u]);
#pragma unroll
                    for (uint32_t u = 0; u < kXReadAhead; ++u) push_pass(mask, acc[u], thr);
                }
                const uint32_t rem = len > it0 ? len - it0 : 0u;  // tests past the window end read other atoms: drop them
                if (rem < kBlock) mask &= ~((1u << (kBlock - rem)) - 1u);
                // Compaction: one round per surviving test of the busiest lane; every round appends <= 64 entries
                // (mask bit 31 - lz <-> test u = lz - (32 - kBlock) <-> chunk record wbase + u)
                const uint32_t tag = lane_tag + wbase - (32u - kBlock);
                STAMP(3)
                while (__any(mask != 0u)) {
                    const uint32_t q0 = __builtin_amdgcn_readfirstlane(qlen);
                    const unsigned long long m = compact_round_x(mask, tag, queue_lds + 4u * q0);
                    uint32_t q1 = q0 + (uint32_t)__popcll(m);
                    if (q1 >= 64) {
                        q1 -= 64;
                        wave_lds_fence();  // lanes read entries other lanes wrote
                        const uint32_t e = w.queue[q1 + lane];
                        wave_lds_fence();
                        STAMP(4)
                        exact_batch_x(prm, w, bl, e, true, slot0, cs, tg, result, lane, wflags, have_res);
                        STAMP(5) STAMP_COUNT(8)
                    }
                    qlen = q1;
                }
                STAMP(4)
            }
            // The prefetched records (and everything older) have landed long ago; saying so HERE, before the drain's store is issued,
            // keeps the compiler from waiting for that store's acknowledgement when the records are committed to LDS above
            // (gfx9 counts loads and stores in one in-order vmcnt).
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            if (qlen) {  // the survivors left at the chunk end: their records are about to be overwritten
                const bool act = lane < qlen;
                wave_lds_fence();
                const uint32_t e = act ? w.queue[lane] : 0u;
                wave_lds_fence();
                STAMP(4)
                exact_batch_x(prm, w, bl, e, act, slot0, cs, tg, result, lane, wflags, have_res);
                STAMP(6) STAMP_COUNT(9)
            }
            k = k2; cs = cs2;
        }
        // hand over to the prefetched task
        if (more && !next_windows) windows_issue(nlo, nhi, cell_start, t_next * 64u + lane, hn.p2.z, t_next * 64u + lane < n_heavy, nx, ny, nzt);
        t = t_next;
        t_next = more ? g_lo + group_waves + __builtin_amdgcn_readfirstlane(grab) : g_hi;
        if (more) { hr = hn; wlo = nlo; whi = nhi; }
    }
#ifdef ARP_STAMP
    st_acc[0] = __builtin_amdgcn_s_memtime() - st_begin;
    if (lane == 0) for (int k = 0; k < 12; k++) atomicAdd(&result[8 + k], st_acc[k]);
#endif
    emit_epilogue(bl, hole_list + blockIdx.x, tg);
}

// single-pass emit + hole fix-up through k_pairs_x: leaves result[0] = number of pairs, out[0..P) contiguous
void launch_emit_x(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof) {
    EmitTarget tg{out, capacity, ws.scratch, ws.scratch_cap, ws.defer_list, ws.defer_cap};
    const uint32_t tasks = (in.n + 63u) / 64u, want = (tasks + kXWaves - 1) / kXWaves;
    const uint32_t nb = want < 1 ? 1 : (want > kXBlocks ? kXBlocks : want);
    if (prof) prof->begin("pairs_emit", st);
    hipLaunchKernelGGL(k_pairs_x, dim3(nb), dim3(kXWaves * 64), 0, st, in, (const GridParams *)ws.grid, (const DevParams *)ws.params,
                       (const uint32_t *)ws.cell_start, ws.sorted, tg, ws.hole_list, ws.task_ctr, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_deferred", st); }
    hipLaunchKernelGGL(k_pairs_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg,
                       ws.hole_list + nb, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_fixup", st); }
    hipLaunchKernelGGL(k_fixup, dim3(256), dim3(kFixThreads), 0, st, (const ulonglong2 *)ws.hole_list, nb + kDeferBlocks, (const GridParams *)ws.grid, tg,
                       ws.result);
    if (prof) prof->end(st);
}
